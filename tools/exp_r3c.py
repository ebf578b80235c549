"""Round-3 experiment C (development aid): (1) decode_code() alone at 32 / 64 / 128 images per call; (2) whole pipeline at
128 images per chain: chains in flight x VQ on its own stream x VQ chunking, A/B/A/B so that drift shows; (3) 64 images per
chain with the fused-norm schedules forced off / default, interleaved.
    python tools/exp_r3c.py
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from llamagen_amd.pipeline import SamplingPipeline
from llamagen_amd.postprocess import to_uint8_hwc

dev = torch.device("cuda:0")
N = 576
skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)


def run(gpt, vq, B, lanes, images, tag, reps=2, **pkw):
    gpt._engine = None
    torch.cuda.empty_cache()
    pipe = SamplingPipeline(gpt, vq, lanes=lanes, **pkw)
    pipe.prepare(B, N, **skw)
    K = max(lanes, images // B)
    K -= K % lanes
    out = []
    for _ in range(reps):
        conds = [torch.randint(0, 1000, (B,), device=dev) for _ in range(K)]
        torch.cuda.synchronize(); t = time.perf_counter()
        pipe.run(conds, N, on_done=(lambda j, i, img: to_uint8_hwc(img)) if vq is not None else None, **skw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        out.append(B * K / dt)
    print(f"{tag:34s} B={B:3d} x {lanes} chains, K={K:2d}, vq={'y' if vq is not None else 'n'}: " + " / ".join(f"{v:6.1f}" for v in out) +
          f" img/s  {pkw}  passes='{os.environ.get('LGEN_PASSES', '')}'", flush=True)
    del pipe
    torch.cuda.empty_cache()


def main():
    torch.set_grad_enabled(False)
    gpt, vq = bench.build_models(dev, 0)
    for B in (32, 64, 128, 32):
        codes = torch.randint(0, 16384, (B, N), device=dev)
        vq.decode_code(codes, [B, 8, 24, 24])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            vq.decode_code(codes, [B, 8, 24, 24])
        e1.record()
        torch.cuda.synchronize()
        print(f"decode_code B={B}: {e0.elapsed_time(e1) / 3:.1f} ms = {e0.elapsed_time(e1) / 3 / B * 32:.1f} ms per 32 images", flush=True)
        del codes
        torch.cuda.empty_cache()
    for rnd in range(2):
        run(gpt, vq, 128, 1, 256, f"R3C r{rnd} 1 chain")
        run(gpt, vq, 128, 1, 256, f"R3C r{rnd} 1 chain chunk32", vq_chunk=32)
        run(gpt, vq, 128, 2, 512, f"R3C r{rnd} 2 chains")
        run(gpt, vq, 128, 2, 512, f"R3C r{rnd} 2 chains chunk32", vq_chunk=32)
        run(gpt, vq, 128, 2, 512, f"R3C r{rnd} 2 chains vq-own-stream", vq_low_priority=True)
        run(gpt, vq, 128, 2, 512, f"R3C r{rnd} 2 chains vq-own chunk32", vq_low_priority=True, vq_chunk=32)
        run(gpt, vq, 128, 3, 768, f"R3C r{rnd} 3 chains chunk32", vq_chunk=32)
    for rnd in range(2):
        for name, spec in (("passes off", "qkv=1,0;w13=1,0;head=1,0"), ("default", ""), ("sweep best", "qkv=2,0;w13=3,0;head=4,0")):
            os.environ["LGEN_PASSES"] = spec
            os.environ["LGEN_TILES"] = "qkv=1,4,8;w13=2,2,8" if name == "sweep best" else ""
            run(gpt, None, 64, 1, 256, f"R3C r{rnd} B64 {name}")
    os.environ["LGEN_PASSES"] = ""
    os.environ["LGEN_TILES"] = ""


if __name__ == "__main__":
    main()
