"""Eager decode-attention launches at a few cache positions (to be run under rocprofv3 --pmc FETCH_SIZE /
WRITE_SIZE): one launch per layer per position, each on its own KV slab."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate
from llamagen_amd import _lib as L

dev = torch.device("cuda:0")
N, B = 576, 32
torch.manual_seed(0)
m = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
m = m.to(device=dev, dtype=torch.bfloat16).eval()
m.setup_caches(2 * B, 1 + N, torch.bfloat16)
e = m._engine
e.k_cache.normal_(0, 1); e.v_cache.normal_(0, 1); e.qbuf.normal_(0, 1)
for pos in (63, 287, 575):
    e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
    for i in range(e.L):
        L.check(e.lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                       2 * B, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, -1, L.stream()), "attn")
    torch.cuda.synchronize()
print("done")
