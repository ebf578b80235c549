"""Do decode-attention launches of different streams overlap their fixed costs?  k streams, each replaying a captured chain of
24 attention launches (own KV slabs), at a few cache positions: time for all streams to finish / (k * 24)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamagen_amd import _lib as L
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.set_grad_enabled(False)
gpt, _ = bench.build_models(dev, 0)
N = 576
engines = []
for i in range(3):
    v = gpt.lane_view()
    v.setup_caches(64, 1 + N, torch.bfloat16)
    e = v._engine
    e.k_cache.normal_(0, 1); e.v_cache.normal_(0, 1)
    engines.append((v, e))
lib = engines[0][1].lib
for pos in (0, 64, 288, 575):
    graphs = []
    for v, e in engines:
        e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
        s = torch.cuda.Stream()
        def chain(e=e):
            for i in range(e.L):
                L.check(lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                             64, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, L.stream()), "attn")
        with torch.cuda.stream(s):
            chain(); s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                chain()
        graphs.append((g, s))
    for k in (1, 2, 3):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                for g, s in graphs[:k]:
                    with torch.cuda.stream(s):
                        g.replay()
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        print(f"pos {pos:3d}: {k} stream(s): {best / (10 * k * 24) * 1e6:6.2f} us per launch-equivalent ({best / 10 * 1e6:7.1f} us per round)", flush=True)
