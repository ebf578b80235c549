"""Smallest possible check of CU-partitioned lane streams: GPT-L decode only (no VQ), 3 lanes, with and without
LGEN_LANE_CU_MASK-style streams -> img/s each.  (tools/sweep_tiles.py is the full experiment.)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models
from llamagen_amd.pipeline import SamplingPipeline

dev = torch.device("cuda:0")
N, B = 576, 32
gpt = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
torch.nn.init.normal_(gpt.output.weight, 0, 0.02)
gpt = gpt.to(device=dev, dtype=torch.bfloat16).eval()
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
ref = None
for part in (True, False):
    gpt._engine = None
    pipe = SamplingPipeline(gpt, None, lanes=3, cu_partition=part)
    pipe.prepare(B, N, **skw)
    conds = [torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(i)).to(dev) for i in range(3)]
    torch.manual_seed(5)
    torch.cuda.synchronize(); t = time.perf_counter()
    out = pipe.run(conds, N, **skw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    ids = torch.stack([o[0] for o in out]).cpu()
    same = "" if ref is None else f" tokens_equal_to_masked_run={bool(torch.equal(ids, ref))}"
    ref = ids if ref is None else ref
    print(f"cu_partition={int(part)}: {3 * B / dt:.1f} img/s (decode only, {dt * 1e3:.0f} ms for 3 batches){same}", flush=True)
