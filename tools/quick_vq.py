"""Quick timing of decode_code on the GPU (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import VQ_models

def main(B=32, lat=24):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
    codes = torch.randint(0, 16384, (B, lat * lat), device=dev)
    from llamagen_amd import _lib as L
    for variant in (3, 0):
      L.lib().lgen_debug_set_igemm_variant(variant)
      print("igemm variant", variant)
      for r in range(3):
            torch.cuda.synchronize(); t = time.time()
            img = vq.decode_code(codes, [B, 8, lat, lat])
            torch.cuda.synchronize(); dt = time.time() - t
            fl = 570.1e9 * B * (lat / 24) ** 2
            print(f"VQ-16 decode B={B} {lat*16}px: {dt*1e3:.1f} ms  {fl/dt/1e12:.1f} TFLOP/s(fp32-equiv)  {B/dt:.1f} img/s  mean={img.mean().item():.4f}", flush=True)
    print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)

if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 32, int(a[1]) if len(a) > 1 else 24)
