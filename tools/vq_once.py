"""One warm decode_code + N timed ones (profiling target: rocprofv3 -- python tools/vq_once.py [B] [reps] [fused])."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(sys.argv) > 3:
    os.environ["LGEN_VQ_FUSED"] = sys.argv[3]
from llamagen_amd import VQ_models
dev = torch.device("cuda:0")
torch.manual_seed(0)
vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
codes = torch.randint(0, 16384, (B, 576), device=dev)
vq.decode_code(codes, [B, 8, 24, 24])
torch.cuda.synchronize(); t = time.time()
for _ in range(reps):
    img = vq.decode_code(codes, [B, 8, 24, 24])
torch.cuda.synchronize()
print(f"decode_code B={B}: {(time.time() - t) / reps * 1e3:.1f} ms")
