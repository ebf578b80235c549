"""Workload for the PMC passes of the round (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a SEPARATE
pass, `--pmc WRITE_SIZE`): (1) a short GPT-L bf16 cfg-4 generate() of LGEN_PMC_B images (default 320 = the bench's ten batches
per chain, 640 rows; the decode-chain GEMM kernels with the bench's tile shapes; 40 tokens keep the serialized,
counter-instrumented run short), (2) the decode attention at cache positions
50 / 300 / 575 on full-size KV slabs, one launch per layer.  tools/pmc_summary.py turns the two outputs into
profiles/r06_pmc.json (LGEN_PMC_TAG), which bench.py quotes as `traffic`."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
N, B = 576, int(os.environ.get("LGEN_PMC_B", "320"))
torch.manual_seed(0)
m = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
torch.nn.init.normal_(m.output.weight, 0, 0.02)
m = m.to(device=dev, dtype=torch.bfloat16).eval()
os.environ["LGEN_NO_GRAPH"] = "1"  # eager launches: every dispatch gets its own counter record
generate(m, torch.randint(0, 1000, (B,), device=dev), 40, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0,
         sample_logits=True)
torch.cuda.synchronize()
m._engine = None
m.setup_caches(2 * B, 1 + N, torch.bfloat16)
e = m._engine
e.k_cache.normal_(0, 1); e.v_cache.normal_(0, 1); e.qbuf.normal_(0, 1)
for pos in (50, 300, 575):   # (round 5: two of the three OFF a 32-key group boundary; 63 / 287 / 575 could not see over-reads past kv_len)
    e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
    for i in range(e.L):
        L.check(e.lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                       2 * B, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, -1, L.stream()), "attn")
    torch.cuda.synchronize()
print("done")
