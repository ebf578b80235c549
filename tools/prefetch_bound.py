"""Upper bound of a weight prefetch for the literal batch-32 call (round 6): the decode-step GEMMs of a 64-row
chain timed (a) over all layers' weights in order (650 MB per step: nothing survives in the 256 MB memory-side cache or the L2s --
what generate() pays), (b) with every layer's launches reading LAYER 0's weights (27 MB: resident in the memory-side cache and mostly in
the L2s -- what a PERFECT, free prefetch one layer ahead could at best deliver), (c) as (a) beside a second stream that streams the
next layer's weights through a copy kernel (a real prefetcher's HBM traffic, without its benefit).  Attention launches of the same
step are timed alongside so that the bound can be priced against the whole step.
    ROWS=64 python tools/prefetch_bound.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402

ROWS = int(os.environ.get("ROWS", "64"))
POS = int(os.environ.get("POS", "288"))
REPS = 40


def gemms(e, layers):
    nw = lambda w: w if e.fuse_norm else None
    e.ssq_parts = e.d // 16
    for i, w in enumerate(layers):
        e.qkv_gemm(i, w, e.hp, nw(w["an"]))
        e.gemm_kind("wo", w)
        e.gemm_kind("w13", w, e.hp, nw(w["fn"]))
        e.gemm_kind("w2", w)
    e.gemm_kind("head", None, e.hp, nw(e.norm_w))


def attn(e):
    for i in range(e.L):
        L.check(e.lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                       e.B2, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, -1, L.stream()), "attn")


def capture(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    return g


def timed(g, reps=REPS, bg=None):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        if bg is not None:
            bg()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.current_stream().synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        torch.cuda.synchronize()
    return best


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    gpt, _ = bench.build_models(dev, 0)
    gpt.setup_caches(ROWS, 1 + 576, torch.bfloat16)
    e = gpt._engine
    e.k_cache.normal_(0, 1); e.v_cache.normal_(0, 1); e.qbuf.normal_(0, 1); e.hp.normal_(0, 1); e.ap.normal_(0, 1); e.gp.normal_(0, 0.1)
    e.ssq.fill_(16.0)
    e.state.copy_(torch.tensor([POS, POS], dtype=torch.int32, device=dev))
    nl = len(e.layers)
    G_cold = capture(lambda: gemms(e, e.layers))
    G_hot = capture(lambda: gemms(e, [e.layers[0]] * nl))
    A = capture(lambda: attn(e))
    t_cold, t_hot, t_att = timed(G_cold), timed(G_hot), timed(A)
    # (c) a prefetcher's traffic beside the cold chain: a second stream copies ~27 MB per layer-time (one step = 650 MB per t_cold)
    side = torch.cuda.Stream()
    dst = torch.empty(8 * 1024 * 1024, dtype=torch.bfloat16, device=dev)
    src = torch.empty(40 * dst.numel(), dtype=torch.bfloat16, device=dev).normal_()   # 671 MB: one decode step's weight stream
    chunks = src.view(-1, dst.numel())

    def background():
        with torch.cuda.stream(side):
            for _ in range(REPS + 4):
                for c in chunks:   # reads 650 MB per pass through a 16 MB window of writes
                    dst.copy_(c)
    t_cold_bg = timed(G_cold, bg=background)
    torch.cuda.synchronize()
    sched = e.gemm_schedule()
    print(f"rows {ROWS}: schedule {sched['wqkv']['family']}")
    print(f"  GEMMs of one decode step, all layers' weights (cold, what generate() pays): {t_cold:8.1f} us")
    print(f"  the same launches on layer 0's weights (cache-resident = a perfect, free prefetch): {t_hot:8.1f} us  ({t_cold / t_hot:4.2f} x)")
    print(f"  cold beside a 650 MB-per-pass copy stream (a prefetcher's traffic):             {t_cold_bg:8.1f} us")
    print(f"  attention launches of the step at position {POS}:                                   {t_att:8.1f} us")
    step = t_cold + t_att
    print(f"  step ~ {step:7.1f} us -> with the perfect prefetch {t_hot + t_att:7.1f} us: {step / (t_hot + t_att):4.2f} x on the GPT phase at best")


if __name__ == "__main__":
    main()
