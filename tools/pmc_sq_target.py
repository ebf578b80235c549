"""Workload of the round-6 SQ counter pass (rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ...): the kernels
of the timed schedule, launched eagerly so that every dispatch gets its own record: (1) a short GPT-L bf16 cfg-4 generate() of 320
images (640 rows: tile GEMM family + persistent attention), (2) the decode attention at positions 50 / 300 / 575 on full-size KV
slabs, (3) one decode_code() of 32 x 24 x 24 codes (conv_fused_kernel and friends).  tools/pmc_sq_summary.py turns the output into
profiles/r06_sq_pmc.csv / .json (`mfma_busy` per kernel family, quoted by bench.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamagen_amd import generate
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
N, B = 576, int(os.environ.get("LGEN_PMC_B", "320"))
torch.set_grad_enabled(False)
gpt, vq = bench.build_models(dev, 0, 2)
os.environ["LGEN_NO_GRAPH"] = "1"
generate(gpt, torch.randint(0, 1000, (B,), device=dev), 12, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0,
         sample_logits=True)
torch.cuda.synchronize()
gpt._engine = None
gpt.setup_caches(2 * B, 1 + N, torch.bfloat16)
e = gpt._engine
e.k_cache.normal_(0, 1); e.v_cache.normal_(0, 1); e.qbuf.normal_(0, 1)
for pos in (50, 300, 575):
    e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
    for i in range(e.L):
        L.check(e.lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                       2 * B, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, -1, L.stream()), "attn")
    torch.cuda.synchronize()
gpt._engine = None
torch.cuda.empty_cache()
idx = torch.randint(0, 16384, (32, 576), device=dev)
for _ in range(2):
    vq.decode_code(idx, [32, 8, 24, 24])
torch.cuda.synchronize()
print("done")
