"""Workload of bench.py's live HBM-traffic passes (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a SEPARATE
pass, `--pmc WRITE_SIZE`): the decode attention -- the bench's dominant kernel -- on full-size GPT-L KV slabs with random
contents, at three cache positions (bench.PMC_POSITIONS), one launch per layer slab for LGEN_PMC_LAYERS layers, LGEN_PMC_ROWS
chain rows.  Only the library and a few buffers are built (no model), so a pass takes seconds."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llamagen_amd import _lib as L
import bench

dev = torch.device("cuda:0")
rows = int(os.environ.get("LGEN_PMC_ROWS", "128"))
nl = int(os.environ.get("LGEN_PMC_LAYERS", "8"))
H, hd, N, T = 16, 64, 576, 1
S8 = (T + N + 7) // 8 * 8
mts = (rows + 15) // 16
mts = (mts + 7) // 8 * 8 if mts > 4 else (4 if mts == 3 else mts)
lib = L.lib()
kc = torch.randn(nl, rows, H, S8, hd, device=dev).to(torch.bfloat16)
vc = torch.randn(nl, rows, H, S8, hd, device=dev).to(torch.bfloat16)
q = torch.randn(mts * 16, H, hd, device=dev).to(torch.bfloat16)
out = torch.zeros(H * hd // 32, mts, 64, 8, dtype=torch.bfloat16, device=dev)
state = torch.zeros(2, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for pos in bench.PMC_POSITIONS(N, T):
    state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
    for i in range(nl):
        L.check(lib.lgen_attn_decode(L.ptr(q), L.ptr(kc[i]), L.ptr(vc[i]), L.ptr(out), L.ptr(state), 0, 0, rows, mts, H, hd, hd, S8, hd,
                                     L.BF16, -1, L.stream()), "attn")
    torch.cuda.synchronize()
print("done")
