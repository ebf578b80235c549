"""Decode-step micro-benchmark (development aid): times the captured hipGraph of one decode step of a
GPT at fixed cache positions, sweeping per-GEMM tile shapes / attention variants / norm fusion, so that
ONE gpurun call answers all tuning questions.

    python tools/ubench_step.py [GPT-L] [B=32] [img=384] [what=all|base|gemm|attn]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402


def time_step(eng, B, sp, pos, reps=24):
    """us per replay of the decode-step graph with the cache position starting at `pos`."""
    st = torch.tensor([pos, pos], dtype=torch.int32, device=eng.dev)
    eng.state.copy_(st)
    eng.decode_step(B, sp)  # eager warm-up (compiles nothing, but first-touch)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        eng.decode_step(B, sp)
    best = 1e9
    for _ in range(3):
        eng.state.copy_(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main(name="GPT-L", B=32, img=384, what="all"):
    dev = torch.device("cuda:0")
    lat = img // 16
    N = lat * lat
    torch.manual_seed(0)
    m = GPT_models[name](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    c = torch.randint(0, 1000, (B,), device=dev)
    kw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    for r in range(2):
        torch.cuda.synchronize()
        t = time.time()
        out = generate(m, c, N, **kw)
        torch.cuda.synchronize()
        dtm = time.time() - t
        print(f"{name} B={B} {img}px: generate {dtm*1e3:.1f} ms ({dtm/N*1e6:.1f} us/step) uniq={out.unique().numel()}", flush=True)
    eng = m._engine
    eng.k_cache.normal_(0, 1)
    eng.v_cache.normal_(0, 1)
    sp = dict(use_cfg=True, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    d, F, V = eng.d, eng.F, eng.V
    P = [8, N // 2, N - 30]

    def row(tag):
        ts = [time_step(eng, B, sp, p) for p in P]
        print(f"{tag:44s} " + "  ".join(f"pos{p}: {t:7.1f} us" for p, t in zip(P, ts)), flush=True)
        return ts

    print("default tiles:", {k: eng._tiles(k, *nk) for k, nk in dict(qkv=(3 * d, d), wo=(d, d), w13=(2 * F, d), w2=(d, F), head=(V, d)).items()})
    # interleaved A/B (run-to-run drift on this box is several %): every config is measured in every round
    configs = {
        "default (attn variant 2)": dict(fuse=True, tiles={}, attn=2),
        "attn variant 1": dict(fuse=True, tiles={}, attn=1),
        "w13(1,2,8)": dict(fuse=True, tiles={"w13": (1, 2, 8)}, attn=2),
        "w13(1,4,8)": dict(fuse=True, tiles={"w13": (1, 4, 8)}, attn=2),
    }
    for rnd in range(2):
        for tag, c in configs.items():
            eng.fuse_norm = c["fuse"]
            eng.prefetch = c.get("pf", False)
            L.lib().lgen_set_attn_variant(c.get("attn", 2))
            eng.tile_override = dict(c["tiles"])
            try:
                row(f"r{rnd} {tag}")
            except Exception as e:
                print(tag, "->", type(e).__name__, e)
    eng.tile_override = {}
    print("max mem GB", torch.cuda.max_memory_allocated() / 2 ** 30)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "GPT-L", int(a[1]) if len(a) > 1 else 32, int(a[2]) if len(a) > 2 else 384,
         a[3] if len(a) > 3 else "all")
