"""Short 3-lane decode run for a rocprofv3 --kernel-trace timeline (development aid)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.ubench_streams import build, capture

dev = torch.device("cuda:0")
N, G, POS, REPS = 576, int(sys.argv[1]) if len(sys.argv) > 1 else 3, 288, 12
streams = [torch.cuda.Stream() for _ in range(G)]
engs = [build("GPT-L", 32, N, dev, s) for s in streams]
gs = [capture(e, 32, POS, s) for (m, e), s in zip(engs, streams)]
for (g, st), (m, e) in zip(gs, engs):
    e.state.copy_(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    for (g, st), s in zip(gs, streams):
        with torch.cuda.stream(s):
            g.replay()
torch.cuda.synchronize()
print("MARK us per batch-step", (time.perf_counter() - t0) / REPS / G * 1e6)
