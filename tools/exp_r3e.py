"""Round-3 experiment E (development aid): does the VQ decoder starve the decode chains of other lanes?  Whole pipeline, 128
images per chain x 3 chains, decode_code in 32-image pieces: default GEMM shapes vs shapes whose workgroups fit beside ONE
resident conv workgroup (<= 156 VGPRs, <= 100 KB LDS), and the decoder confined to a CU subset (masked stream).
    python tools/exp_r3e.py
"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tools.exp_r3c import run  # noqa: E402

LOW = dict(LGEN_TILES="w13=2,2,8;wo=2,1,8;w2=2,2,8;head=1,4,8", LGEN_PASSES="qkv=3,0;w13=6,0;head=8,0")


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    gpt, vq = bench.build_models(dev, 0)
    for rnd in range(2):
        for name, env, pkw in (("default", {}, {}), ("low-reg tiles", LOW, {}), ("default vq on 224 CUs", {}, dict(vq_cus=224)),
                               ("low-reg vq on 224 CUs", LOW, dict(vq_cus=224)), ("low-reg vq on 192 CUs", LOW, dict(vq_cus=192))):
            os.environ["LGEN_TILES"] = env.get("LGEN_TILES", "")
            os.environ["LGEN_PASSES"] = env.get("LGEN_PASSES", "")
            run(gpt, vq, 128, 3, 768, f"R3E r{rnd} {name}", vq_chunk=32, **pkw)
        os.environ["LGEN_TILES"] = LOW["LGEN_TILES"]; os.environ["LGEN_PASSES"] = LOW["LGEN_PASSES"]
        run(gpt, None, 128, 3, 768, f"R3E r{rnd} low-reg decode only")
    os.environ["LGEN_TILES"] = ""; os.environ["LGEN_PASSES"] = ""


if __name__ == "__main__":
    main()
