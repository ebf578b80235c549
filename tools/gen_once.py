"""One generate() at a given batch (profiling target): python tools/gen_once.py [B] [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from llamagen_amd import generate
B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N = int(sys.argv[2]) if len(sys.argv) > 2 else 576
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); torch.set_grad_enabled(False)
gpt, _ = bench.build_models(dev, 0)
c = torch.randint(0, 1000, (B,), device=dev)
kw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
generate(gpt, c, N, **kw)
torch.cuda.synchronize(); t = time.time()
generate(gpt, c, N, **kw)
torch.cuda.synchronize(); print(f"generate B={B} N={N}: {(time.time() - t) * 1e3:.1f} ms")
