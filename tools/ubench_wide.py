"""images/s of the config-2 hot path (GPT-L 384 px, cfg 4.0, top-k 2000 + VQ decode) as a function of the rows
per decode chain (B images = B/32 reference batches carried by ONE chain of launches) and of the chains in flight.
Development aid for the "wide step" schedule; the bench line stays bench.py."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, VQ_models
from llamagen_amd.pipeline import SamplingPipeline

dev = torch.device("cuda:0")
torch.manual_seed(0)
N = 576
gpt = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
torch.nn.init.normal_(gpt.output.weight, 0, 0.02)
gpt = gpt.to(device=dev, dtype=torch.bfloat16).eval()
vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
grid = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32, 1), (32, 3), (64, 1), (64, 2), (96, 1), (96, 2), (128, 1)]
for B, lanes in grid:
    pipe = SamplingPipeline(gpt, vq, lanes=lanes)
    pipe.prepare(B, N, **skw)
    K = max(2 * lanes, 192 // B)
    K -= K % lanes
    conds = [torch.randint(0, 1000, (B,), device=dev) for _ in range(K)]
    torch.cuda.synchronize(); t = time.perf_counter()
    out = pipe.run(conds, N, **skw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    assert all(torch.isfinite(img).all() for _, img in out)
    print(f"B={B} lanes={lanes} K={K}: {B * K / dt:.1f} img/s ({dt / K * 1e3:.0f} ms per chain-batch)", flush=True)
    del pipe, out
    torch.cuda.empty_cache()
