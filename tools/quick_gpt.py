"""Quick timing of generate() on the GPU (development aid, not the contract bench)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate  # noqa: E402


def main(name="GPT-L", B=32, img=384, cfg=4.0, reps=3, prec="bf16"):
    dev = torch.device("cuda:0")
    lat = img // 16
    dt = {"bf16": torch.bfloat16, "fp32": torch.float32}[prec]
    torch.manual_seed(0)
    m = GPT_models[name](vocab_size=16384, block_size=lat * lat, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=dt).eval()
    c = torch.randint(0, 1000, (B,), device=dev)
    kw = dict(cfg_scale=cfg, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    for r in range(reps):
        torch.cuda.synchronize()
        t = time.time()
        out = generate(m, c, lat * lat, **kw)
        torch.cuda.synchronize()
        dtm = time.time() - t
        print(f"{name} B={B} {img}px cfg={cfg} {prec}: generate {dtm*1e3:.1f} ms  ({dtm/ (lat*lat) * 1e6:.1f} us/step, {B/dtm:.1f} img/s GPT only) "
              f"uniq={out.unique().numel()}", flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "GPT-L", int(a[1]) if len(a) > 1 else 32, int(a[2]) if len(a) > 2 else 384,
         float(a[3]) if len(a) > 3 else 4.0, 3, a[4] if len(a) > 4 else "bf16")
