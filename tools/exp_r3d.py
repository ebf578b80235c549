"""Round-3 experiment D (development aid): 512-row chains (8 batches of 32 per chain) against 256-row chains, decode only and
whole pipeline (decode_code in 32-image pieces).
    python tools/exp_r3d.py
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from llamagen_amd.pipeline import SamplingPipeline
from llamagen_amd.postprocess import to_uint8_hwc
from tools.exp_r3c import run  # noqa: E402  (same measurement loop)


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    gpt, vq = bench.build_models(dev, 0)
    for B, lanes, images in [(256, 1, 512), (256, 2, 1024), (128, 2, 1024), (128, 3, 768)]:
        run(gpt, None, B, lanes, images, f"R3D decode-only")
    for B, lanes, images in [(256, 1, 512), (256, 2, 1024), (128, 3, 768)]:
        run(gpt, vq, B, lanes, images, f"R3D with vq chunk32", vq_chunk=32)


if __name__ == "__main__":
    main()
