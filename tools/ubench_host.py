"""Host cost of replaying the decode-step graph (development aid)."""
import os, sys, time, threading
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ubench_streams import build, capture  # noqa: E402

dev = torch.device("cuda:0")
N = 576
streams = [torch.cuda.Stream() for _ in range(2)]
engs = [build("GPT-L", 32, N, dev, s) for s in streams]
gs = [capture(e, 32, 8, s) for (m, e), s in zip(engs, streams)]
g, st = gs[0]
e = engs[0][1]
for reps in (1, 20):
    e.state.copy_(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        for _ in range(reps):
            g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"reps {reps}: host enqueue {1e6*(t1-t0)/reps:.1f} us/replay, total {1e6*(t2-t0)/reps:.1f} us/replay", flush=True)

def worker(i, reps):
    g, st = gs[i]
    with torch.cuda.stream(streams[i]):
        for _ in range(reps):
            g.replay()

for (g_, st_), (m_, e_) in zip(gs, engs):
    e_.state.copy_(st_)
torch.cuda.synchronize()
t0 = time.perf_counter()
ths = [threading.Thread(target=worker, args=(i, 20)) for i in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"2 threads x 20 replays: host {1e6*(t1-t0)/20:.1f} us per pair, total {1e6*(t2-t0)/20:.1f} us per pair", flush=True)
