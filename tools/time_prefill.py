"""t2i prefix prefill timing: batched vs position-by-position (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate
dev = torch.device("cuda:0")
B, N = 16, 1024
torch.manual_seed(0)
m = GPT_models["GPT-XL"](vocab_size=16384, block_size=N, model_type="t2i", cls_token_num=120, caption_dim=2048)
torch.nn.init.normal_(m.output.weight, 0, 0.02)
m = m.to(device=dev, dtype=torch.bfloat16).eval()
c = torch.randn(B, 120, 2048, device=dev, dtype=torch.bfloat16)
masks = torch.ones(B, 120, dtype=torch.int64, device=dev)
for mode in ("1", "0", "1", "0"):
    os.environ["LGEN_SEQ_PREFILL"] = mode
    for n in (4, 4):
        torch.cuda.synchronize(); t = time.time()
        generate(m, c, n, emb_masks=masks, cfg_scale=7.5, top_k=1000, temperature=1.0, top_p=1.0, sample_logits=True)
        torch.cuda.synchronize(); dt = time.time() - t
    print("sequential" if mode == "1" else "batched   ", f"prefill + 3 decode steps: {dt*1e3:.1f} ms", flush=True)
