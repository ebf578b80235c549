"""Do the kernels of two decode chains really run side by side?  (development aid, round 4)

Two engines (shared weights, own KV slabs) on two HIP streams.  Graph A = the 24 attention launches of a decode step at position P,
graph G = the 24 x 4 GEMM launches + lm_head of a decode step, graph V = one decode_code() of 32 images.  Each graph is timed alone
and in pairs on two streams (A|G, A|A, G|G, A|V, G|V): perfect overlap = max of the two, none = their sum.
    ROWS=256 POS=300 python tools/overlap_probe.py
"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402

ROWS = int(os.environ.get("ROWS", "256"))
POS = int(os.environ.get("POS", "300"))
REPS = 30
ATTN_VARIANT = int(os.environ.get("ATTN_VARIANT", "-1"))   # lgen_attn_decode variant of graph A (-1 = the library's choice)


def attn_only(e):
    lib, st = e.lib, L.stream()
    for i in range(e.L):
        L.check(lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                     e.B2, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, ATTN_VARIANT, st), "attn")


def gemms_only(e):
    nw = lambda w: w if e.fuse_norm else None
    e.ssq_parts = e.d // 16
    for i, w in enumerate(e.layers):
        e.qkv_gemm(i, w, e.hp, nw(w["an"]))
        e.gemm_kind("wo", w)
        e.gemm_kind("w13", w, e.hp, nw(w["fn"]))
        e.gemm_kind("w2", w)
    e.gemm_kind("head", None, e.hp, nw(e.norm_w))


def capture(fn, stream):
    with torch.cuda.stream(stream):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    torch.cuda.synchronize()
    return g


def timed(pairs):
    """pairs: [(graph, stream), ...] replayed REPS times each, interleaved; us per replay round"""
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(REPS):
            for g, s in pairs:
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / REPS * 1e6)
    return best


def timed_under(fg, bg, t_fg, t_bg):
    """us per replay of graph `fg` (events on its stream around REPS replays) while `bg` replays back to back on the other stream
    for at least twice as long: how far a chain's kernels stretch beside the other chain's."""
    (g1, s1), (g2, s2) = fg, bg
    nbg = max(REPS, int(3.0 * REPS * max(t_fg, 1.0) / max(t_bg, 1.0)) + REPS)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s2):
        for _ in range(REPS // 2):
            g2.replay()
    with torch.cuda.stream(s1):
        e0.record()
    k = 0
    for i in range(REPS):
        with torch.cuda.stream(s1):
            g1.replay()
        while k < (i + 1) * nbg // REPS:
            with torch.cuda.stream(s2):
                g2.replay()
            k += 1
    with torch.cuda.stream(s1):
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    gpt, vq = bench.build_models(dev, 0)
    N = 576
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    engs = []
    for view in (gpt, gpt.lane_view()):
        view.setup_caches(ROWS, 1 + N, torch.bfloat16)
        e = view._engine
        e.k_cache.normal_(0, 1)
        e.v_cache.normal_(0, 1)
        e.qbuf.normal_(0, 1)
        e.hp.normal_(0, 1)
        e.ap.normal_(0, 1)
        e.gp.normal_(0, 0.1)
        e.ssq.fill_(16.0)
        e.state.copy_(torch.tensor([POS, POS], dtype=torch.int32, device=dev))
        engs.append((view, e))
    (_, e1), (_, e2) = engs
    codes = torch.randint(0, 16384, (32, N), device=dev)
    A1, A2 = capture(lambda: attn_only(e1), s1), capture(lambda: attn_only(e2), s2)
    G1, G2 = capture(lambda: gemms_only(e1), s1), capture(lambda: gemms_only(e2), s2)

    def vqd():
        vq.decode_code(codes, (32, 8, 24, 24))
    with torch.cuda.stream(s2):
        vqd()
    torch.cuda.synchronize()
    tA, tG = timed([(A1, s1)]), timed([(G1, s1)])
    print(f"rows {ROWS} pos {POS}: A alone {tA:8.1f} us   G alone {tG:8.1f} us   (tile {e1._tile_shape('qkv')})", flush=True)
    for name, pairs, parts in (("A|G", [(A1, s1), (G2, s2)], (tA, tG)), ("A|A", [(A1, s1), (A2, s2)], (tA, tA)),
                               ("G|G", [(G1, s1), (G2, s2)], (tG, tG))):
        t = timed(pairs)
        print(f"  {name}: {t:8.1f} us   sum {sum(parts):8.1f}  max {max(parts):8.1f}  -> overlap {(sum(parts) - t) / min(parts):5.2f} "
              f"(1 = the shorter one fully hidden)", flush=True)
    tGA, tAG = timed_under((G2, s2), (A1, s1), tG, tA), timed_under((A1, s1), (G2, s2), tA, tG)
    print(f"  G under A: {tGA:8.1f} us per GEMM graph ({tGA / tG:4.2f} x alone)   A under G: {tAG:8.1f} us per attention graph ({tAG / tA:4.2f} x alone)",
          flush=True)
    if os.environ.get("NO_V") == "1":
        return
    # decoder beside a chain: REPS x A (or G) on s1 while one decode_code() of 32 images runs on s2
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        vqd()
    torch.cuda.synchronize()
    tV = (time.perf_counter() - t0) * 1e6
    for name, g, tg in (("A", A1, tA), ("G", G1, tG)):
        reps = max(1, int(tV / tg))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(s2):
            vqd()
        for _ in range(reps):
            with torch.cuda.stream(s1):
                g.replay()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) * 1e6
        print(f"  V|{name}: V alone {tV:8.1f} us, {reps} x {name} alone {reps * tg:8.1f} us, together {t:8.1f} us -> overlap "
              f"{(tV + reps * tg - t) / min(tV, reps * tg):5.2f}", flush=True)


if __name__ == "__main__":
    main()
