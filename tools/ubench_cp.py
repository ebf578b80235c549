"""Is kernel dispatch serialised across HIP streams?  G streams, each replaying a graph of 500 trivial
dependent kernels: per-kernel time of all streams together (development aid)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.lib()
N = 500
for G in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(G)]
    states = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(G)]
    graphs = []
    for s, stt in zip(streams, states):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                lib.lgen_advance_state(stt.data_ptr(), L.stream())
        graphs.append(g)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 4)
    print(f"{G} streams x {N} trivial dependent kernels: {best*1e6/N:.2f} us per kernel-slot ({best*1e6/(N*G):.2f} us per kernel overall)", flush=True)
