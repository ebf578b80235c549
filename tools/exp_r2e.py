"""Round-2 experiment batch E (development aid; results are quoted in DESIGN.md section 5b).
Whole pipeline (generate + decode_code + uint8), GPT-L 384 px, cfg 4.0: images per chain x chains in flight, and GEMM
workgroup shapes for the 128-row chain (LGEN_TILES).  One process, so that the configurations see the same chip / clocks;
every configuration runs twice.
    python tools/exp_r2e.py [grid|tiles ...]
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


def run(gpt, vq, B, lanes, images=384, tag="", reps=2):
    gpt._engine = None
    pipe = SamplingPipeline(gpt, vq, lanes=lanes)
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
    print(f"{tag} B={B:3d} x {lanes} chains, K={K:2d}, vq={'y' if vq is not None else 'n'}: " + " / ".join(f"{v:6.1f}" for v in out) +
          f" img/s   tiles='{os.environ.get('LGEN_TILES', '')}'", flush=True)
    del pipe
    torch.cuda.empty_cache()


def main():
    which = sys.argv[1:] or ["grid", "tiles"]
    torch.set_grad_enabled(False)
    gpt, vq = bench.build_models(dev, 0)
    if "grid" in which:
        for B, lanes in [(32, 3), (64, 1), (64, 2), (64, 3), (96, 2), (32, 1), (32, 3)]:
            run(gpt, vq, B, lanes, tag="E5")
    if "wide" in which:
        for B, lanes in [(128, 1), (128, 2), (64, 3)]:
            run(gpt, vq, B, lanes, images=512, tag="E7")
    if "tiles" in which:
        # (round 2, after the statistics of wide m-tiles were fixed to one L2 round trip; qkv=2,4,8 is the default at 128 rows)
        sets = {"default": "", "w22": "w13=2,2,8", "h22": "head=2,2,8", "q22": "qkv=2,2,8", "q14": "qkv=1,4,8", "w14": "w13=1,4,8",
                "w22_q22": "w13=2,2,8;qkv=2,2,8"}
        for name, spec in sets.items():
            os.environ["LGEN_TILES"] = spec
            try:
                run(gpt, None, 64, 3, images=384, tag=f"E6 {name:13s}", reps=2)
                run(gpt, None, 64, 1, images=128, tag=f"E6 {name:13s}", reps=2)
            except Exception as ex:  # an uninstantiated shape: keep going
                print(f"E6 {name}: failed: {ex!r}", flush=True)
        os.environ["LGEN_TILES"] = ""


if __name__ == "__main__":
    main()
