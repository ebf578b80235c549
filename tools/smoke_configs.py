"""Run BASELINE.json configs 3-5 (per-GPU shapes) once each on the GPU: crash / shape / finiteness check
and a coarse timing (development aid; the bench line is config 2)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, VQ_models, generate

dev = torch.device("cuda:0")
vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()

def run(name, B, img, model_type="c2i", cfg=4.0, top_k=2000, reps=2):
    lat = img // 16
    N = lat * lat
    torch.manual_seed(0)
    kw = dict(vocab_size=16384, block_size=N, model_type=model_type)
    if model_type == "c2i":
        kw.update(num_classes=1000, cls_token_num=1)
    else:
        kw.update(cls_token_num=120, caption_dim=2048)
    m = GPT_models[name](**kw)
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    if model_type == "c2i":
        c, masks = torch.randint(0, 1000, (B,), device=dev), None
    else:
        c = torch.randn(B, 120, 2048, device=dev, dtype=torch.bfloat16)
        lens = torch.randint(5, 121, (B,))
        masks = torch.zeros(B, 120, dtype=torch.int64, device=dev)
        for b in range(B):
            masks[b, 120 - int(lens[b]):] = 1
        c = c * masks[:, :, None]
    for r in range(reps):
        torch.cuda.synchronize(); t = time.time()
        idx = generate(m, c, N, emb_masks=masks, cfg_scale=cfg, cfg_interval=-1, temperature=1.0, top_k=top_k, top_p=1.0, sample_logits=True)
        torch.cuda.synchronize(); t1 = time.time()
        img_t = vq.decode_code(idx, [B, 8, lat, lat])
        torch.cuda.synchronize(); t2 = time.time()
        assert idx.shape == (B, N) and img_t.shape == (B, 3, img, img) and torch.isfinite(img_t).all()
        print(f"{name} {model_type} B={B} {img}px: generate {1e3*(t1-t):.0f} ms, decode {1e3*(t2-t1):.0f} ms, {B/(t2-t):.1f} img/s, "
              f"uniq tokens {idx.unique().numel()}, fused_norm={m._engine.fuse_norm}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB", flush=True)
    del m
    torch.cuda.empty_cache()

which = sys.argv[1:] or ["c3", "c4", "c5"]
if "c3" in which: run("GPT-XXL", 32, 384)
if "c4" in which: run("GPT-3B", 64, 384)
if "c5" in which: run("GPT-XL", 16, 512, model_type="t2i", cfg=7.5, top_k=1000)
