"""Round-3 experiment A (development aid): decode-only and whole-pipeline throughput for images-per-chain x chains in
flight, with the default GEMM workgroup shapes and with the per-row-count best shapes found by tools/gemm_sweep.py
(gpurun_out/gemm_sweep.json).  One process; every configuration runs twice.
    python tools/exp_r3a.py
"""
import json, os, sys, time
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
    torch.cuda.empty_cache()
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
          f" img/s   tiles='{os.environ.get('LGEN_TILES', '')}' passes='{os.environ.get('LGEN_PASSES', '')}'", flush=True)
    del pipe
    torch.cuda.empty_cache()


def main():
    torch.set_grad_enabled(False)
    gpt, vq = bench.build_models(dev, 0)
    best = {}
    try:
        best = {int(k): v for k, v in json.load(open(os.path.join(ROOT, "gpurun_out", "gemm_sweep.json")))["best"].items()}
    except Exception as ex:  # noqa: BLE001
        print("no sweep result:", ex)
    for B, lanes, images in [(32, 1, 64), (64, 1, 128), (128, 1, 256), (128, 2, 512), (64, 3, 384)]:
        for name, spec in (("default", None), ("best", best.get(2 * B))):
            if name == "best" and not spec:
                continue
            os.environ["LGEN_TILES"] = spec["tiles"] if spec else ""
            os.environ["LGEN_PASSES"] = spec["passes"] if spec else ""
            try:
                run(gpt, None, B, lanes, images=images, tag=f"R3A {name:7s}")
            except Exception as ex:  # noqa: BLE001
                print(f"R3A {name} B={B} x {lanes}: failed: {ex!r}", flush=True)
    for B, lanes, images in [(128, 1, 256), (128, 2, 512), (64, 3, 384)]:
        spec = best.get(2 * B)
        os.environ["LGEN_TILES"] = spec["tiles"] if spec else ""
        os.environ["LGEN_PASSES"] = spec["passes"] if spec else ""
        run(gpt, vq, B, lanes, images=images, tag="R3A best+vq")
    os.environ["LGEN_TILES"] = ""
    os.environ["LGEN_PASSES"] = ""


if __name__ == "__main__":
    main()
