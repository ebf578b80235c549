"""Multi-lane throughput of config 2 under different GEMM workgroup shapes (engine LGEN_TILES hook).

Hypothesis behind it: with 3 batches in flight the decode GEMMs take turns on the chip,
because each launch spreads 176-512 workgroups of 512 threads over all 256 CUs; fewer, fatter workgroups per
GEMM (64-96) would let the three lanes' latency-bound kernels run side by side.

    python tools/sweep_tiles.py            # on a GPU: img/s for every candidate, 3 lanes (shared chip and
                                           # CU-partitioned lane streams) and 1 lane
    python tools/sweep_tiles.py --check    # no GPU: only validates that the library accepts every candidate
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CANDIDATES = {
    "default": "",
    "fat_res": "wo=4,1,8;w2=4,1,8",                                            # 64 WGs instead of 256
    "fat_qkv": "qkv=2,4,8",                                                    # 96 instead of 192
    "fat_all": "qkv=2,4,8;wo=4,1,8;w13=4,4,8;w2=4,1,8;head=4,4,8",             # 96 / 64 / 88 / 64 / 256
    "fat_all_res2": "qkv=2,4,8;wo=2,1,8;w13=4,4,8;w2=2,1,8;head=4,4,8",        # wo / w2 at 128
    "mid": "wo=2,1,8;w2=2,1,8",
}


def check():
    from llamagen_amd import _lib as L
    lib = L.lib()
    d, H, F, mts, B2, kch = 1024, 16, 2816, 4, 64, 32
    ok = True
    for name, spec in CANDIDATES.items():
        for item in filter(None, spec.split(";")):
            kind, _, val = item.partition("=")
            mt, nt, kw = (int(v) for v in val.split(","))
            nw = 8 if kind in ("qkv", "w13", "head") else 0
            if kind == "qkv":
                rc = lib.lgen_gemm_qkv_rope(8, 8, 8, 8, 8, 8, 8, B2, mts, d, H, 64, 64, 584, 0, L.BF16, mt, nt, kw, nw, 8, kch, 1e-5, 0)
            else:
                N, K, epi = {"wo": (d, d, L.EPI_RES), "w2": (d, F, L.EPI_RES), "w13": (2 * F, d, L.EPI_SWIGLU),
                             "head": (16384, d, L.EPI_ROWS)}[kind]
                rc = lib.lgen_gemm(8, 8, 8, B2, mts, N, K, epi, L.BF16, mt, nt, kw, nw, 8 if nw else 0, kch, 1e-5, 0, 0)
            if rc in (-1, -2):
                ok = False
            print(f"{name:14s} {item:12s} -> rc {rc} ({'rejected' if rc in (-1, -2) else 'accepted'})")
    return ok


def main():
    from llamagen_amd import GPT_models, VQ_models
    from llamagen_amd.pipeline import SamplingPipeline
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, B = 576, 32
    gpt = GPT_models["GPT-L"](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(gpt.output.weight, 0, 0.02)
    gpt = gpt.to(device=dev, dtype=torch.bfloat16).eval()
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
    skw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    names = sys.argv[1:] or list(CANDIDATES)
    for name in names:
        os.environ["LGEN_TILES"] = CANDIDATES[name]
        for lanes, part in ((3, False), (3, True), (2, True), (4, True), (1, False)):
            gpt._engine = None  # engines read LGEN_TILES when they are built
            pipe = SamplingPipeline(gpt, vq, lanes=lanes, cu_partition=part)
            pipe.prepare(B, N, **skw)
            K = {1: 2, 2: 4, 3: 6, 4: 8}[lanes]
            conds = [torch.randint(0, 1000, (B,), device=dev) for _ in range(K)]
            torch.cuda.synchronize(); t = time.perf_counter()
            pipe.run(conds, N, **skw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            print(f"{name:14s} lanes={lanes} cu_partition={int(part)}: {B * K / dt:6.1f} img/s   [{CANDIDATES[name] or 'engine heuristics'}]", flush=True)
            del pipe
            torch.cuda.empty_cache()


if __name__ == "__main__":
    if "--check" in sys.argv:
        sys.exit(0 if check() else 1)
    main()
