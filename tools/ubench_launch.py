"""Measure per-kernel fixed costs on this box: chains of trivial kernels, eager vs hipGraph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
lib = L.lib()
state = torch.zeros(2, dtype=torch.int32, device=dev)
N = 1000

def chain():
    st = L.stream()
    for _ in range(N):
        lib.lgen_advance_state(state.data_ptr(), st)

def timeit(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); ts.append(time.time() - t)
    return min(ts)

print("eager 1000 x advance_state: %.2f us/kernel" % (timeit(chain) / N * 1e6))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
print("graph 1000 x advance_state: %.2f us/kernel" % (timeit(g.replay) / N * 1e6))
# torch elementwise op chain for comparison
x = torch.zeros(64, device=dev)
def tchain():
    for _ in range(N): x.add_(1.0)
print("eager 1000 x torch add_: %.2f us/kernel" % (timeit(tchain) / N * 1e6))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    tchain()
print("graph 1000 x torch add_: %.2f us/kernel" % (timeit(g2.replay) / N * 1e6))
os.system("rocm-smi --showclocks 2>/dev/null | head -30")
