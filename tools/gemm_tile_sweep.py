"""Big-M tile GEMM family (csrc/gemm_tile.hip) against the skinny kernels: correctness + timing sweep (development aid, round 4).

For each chain width (rows) and each GEMM kind of the decode step, every candidate workgroup shape (wm, wn, mtv, ntv, kb, stages)
is (1) compared with the skinny kernel's output on the same operands (same math up to fp32 accumulation order: reports the
fraction of differing bf16 elements and the largest difference in bf16 ulps of the value) and (2) timed as a captured chain over
all layers' weights (nothing cache-resident), next to the skinny kernel with the engine's current default shape.
    python tools/gemm_tile_sweep.py [model] [rows ...]      (default GPT-L 256 128)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402


def timed_graph(fn, launches, reps=6):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * launches))
    return best


def ulp_diff(a, b):
    """(fraction of elements that differ, max |a - b| in bf16 ulps of max(|a|, |b|, rms): a value near zero is the sum of
    O(rms)-sized terms, its absolute error is what the accumulation order moves)"""
    af, bf = a.float(), b.float()
    diff = (af - bf).abs()
    rms = float(bf.pow(2).mean().sqrt().clamp_min(1e-30))
    scale = torch.maximum(af.abs(), bf.abs()).clamp_min(rms)
    ulp = torch.exp2(torch.floor(torch.log2(scale)) - 7)
    return float((diff > 0).float().mean()), float((diff / ulp).max())


NORM_SHAPES = [(4, 1, 1, 3, 4, 4, 4), (4, 1, 1, 4, 4, 4, 4), (4, 1, 1, 6, 2, 4, 4), (4, 1, 1, 6, 4, 3, 4), (4, 1, 1, 8, 2, 4, 4),
               (4, 1, 2, 4, 2, 4, 4), (4, 1, 2, 6, 2, 4, 4), (4, 1, 2, 8, 2, 4, 4), (4, 1, 1, 2, 4, 4, 4), (4, 1, 1, 3, 4, 4, 0),
               (8, 1, 1, 8, 2, 4, 4), (8, 1, 1, 6, 2, 4, 4), (8, 1, 1, 4, 2, 4, 4), (4, 1, 2, 10, 2, 4, 4)]
PLAIN_SHAPES = [(2, 2, 1, 1, 4, 4, 4), (2, 2, 1, 2, 4, 4, 4), (2, 2, 2, 1, 4, 4, 4), (2, 2, 2, 2, 4, 4, 4), (2, 2, 2, 2, 2, 4, 4),
                (4, 1, 1, 2, 4, 4, 4), (2, 2, 1, 1, 4, 4, 0), (2, 2, 4, 1, 4, 4, 4), (2, 2, 4, 2, 2, 4, 4)]


def sweep(m, rows, out):
    N = 576
    m._engine = None
    torch.cuda.empty_cache()
    m.setup_caches(rows, 1 + N, torch.bfloat16)
    e = m._engine
    lib, dt, M, mts = e.lib, e.dt, e.B2, e.MTs
    d, F, H, hd, hdp, S8, V = e.d, e.F, e.H, e.hd, e.hdp, e.S8, e.V
    torch.manual_seed(1)
    e.hp.normal_(0, 1)
    e.ap.normal_(0, 1)
    e.gp.normal_(0, 1)
    hp0 = e.hp.clone()
    # statistics that belong to hp0 (what the producer's epilogue would have written)
    L.check(lib.lgen_ssq_pack(L.ptr(e.hp), L.ptr(e.ssq), mts, d, dt, L.stream()), "ssq_pack")
    ssq0 = e.ssq.clone()
    e.ssq_parts = d // 16
    # the tile family's consumers get the rows the engine gives them: grouped once (lgen_ssq_group4, round 6) where d / 16 is not a
    # multiple of 16 (GPT-3B), else as they are; SWEEP_GROUP4=0 times the un-grouped rows
    ssq_t, parts_t = ssq0, e.ssq_parts
    if (parts_t % 16 or parts_t > 128) and os.environ.get("SWEEP_GROUP4", "1") != "0":
        ssq_t = torch.zeros_like(ssq0)
        L.check(lib.lgen_ssq_group4(L.ptr(ssq0), L.ptr(ssq_t), mts * 16, parts_t, L.stream()), "ssq_group4")
        parts_t = 4
    e.state.zero_()
    e.state[0] = 7
    nl = len(e.layers)
    ssq_scratch = torch.zeros_like(e.ssq)

    def qkv_skinny(w, tl, sc):
        L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(e.hp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                       L.ptr(e.freqs_cis), L.ptr(e.state), M, mts, d, H, hd, hdp, S8, e.kvs, dt, tl[0], tl[1], tl[2],
                                       L.ptr(w["an"]), L.ptr(ssq0), e.ssq_parts, e.eps, sc[0], L.stream()), "qkv")

    def qkv_tile(w, s):
        return lib.lgen_gemm_qkv_rope_tile(L.ptr(w["wqkv"]), L.ptr(e.hp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                           L.ptr(e.freqs_cis), L.ptr(e.state), M, mts, d, H, hd, hdp, S8, e.kvs, dt, *s,
                                           L.ptr(w["an"]), L.ptr(ssq_t), parts_t, e.eps, L.stream())

    def gemm_tile(wp, xp, o, Nn, K, epi, s, nw=None, ssq_out=None):
        return lib.lgen_gemm_tile(L.ptr(wp), L.ptr(xp), L.ptr(o), M, mts, Nn, K, epi, dt, *s, L.ptr(nw),
                                  L.ptr(ssq_t) if nw is not None else 0, parts_t, e.eps, L.ptr(ssq_out), L.stream())

    # kind -> (skinny(w, tiles, sched), tile(w, shape) -> rc, outputs(), reset(), weights list, shapes, N, K)
    def outs_qkv():
        return [e.qbuf.clone(), e.k_cache[0].clone(), e.v_cache[0].clone()]

    def reset_qkv():
        e.qbuf.zero_(); e.k_cache[0].zero_(); e.v_cache[0].zero_()

    def reset_res():
        e.hp.copy_(hp0); ssq_scratch.zero_()

    head_w = [dict(out=e.out_w, nw=e.norm_w)]
    kinds = {
        "qkv": (lambda w, tl, sc: qkv_skinny(w, tl, sc), qkv_tile, outs_qkv, reset_qkv, e.layers, NORM_SHAPES, 3 * d, d),
        "w13": (lambda w, tl, sc: e.gemm(w["w13"], e.hp, e.gp, M, mts, 2 * F, d, L.EPI_SWIGLU, tl, norm_w=w["fn"], sched=sc),
                lambda w, s: gemm_tile(w["w13"], e.hp, e.gp, 2 * F, d, L.EPI_SWIGLU, s, nw=w["fn"]),
                lambda: [e.gp.clone()], lambda: e.gp.zero_(), e.layers, NORM_SHAPES, 2 * F, d),
        "head": (lambda w, tl, sc: e.gemm(w["out"], e.hp, e.logits, M, mts, V, d, L.EPI_ROWS, tl, norm_w=w["nw"], sched=sc),
                 lambda w, s: gemm_tile(w["out"], e.hp, e.logits, V, d, L.EPI_ROWS, s, nw=w["nw"]),
                 lambda: [e.logits[:M].clone()], lambda: e.logits.zero_(), head_w * 4, NORM_SHAPES, V, d),
        "wo": (lambda w, tl, sc: e.gemm(w["wo"], e.ap, e.hp, M, mts, d, d, L.EPI_RES, tl, ssq_out=ssq_scratch),
               lambda w, s: gemm_tile(w["wo"], e.ap, e.hp, d, d, L.EPI_RES, s, ssq_out=ssq_scratch),
               lambda: [e.hp.clone(), ssq_scratch[:, :d // 16].clone()], reset_res, e.layers, PLAIN_SHAPES, d, d),
        "w2": (lambda w, tl, sc: e.gemm(w["w2"], e.gp, e.hp, M, mts, d, F, L.EPI_RES, tl, ssq_out=ssq_scratch),
               lambda w, s: gemm_tile(w["w2"], e.gp, e.hp, d, F, L.EPI_RES, s, ssq_out=ssq_scratch),
               lambda: [e.hp.clone(), ssq_scratch[:, :d // 16].clone()], reset_res, e.layers, PLAIN_SHAPES, d, F),
    }
    only = os.environ.get("LGEN_SWEEP_KINDS")
    for kind, (skinny, tile, outs, reset, ws, shapes, Nn, K) in kinds.items():
        if only and kind not in only.split(","):
            continue
        tl = e._tiles(kind, Nn, K)
        sc = e._passes(kind, Nn, tl)
        if kind == "w13":   # the activations of the SwiGLU input are the residual stream
            e.hp.copy_(hp0)
        reset()
        skinny(ws[0], tl, sc)
        ref = outs()
        if kind in ("wo", "w2"):
            e.hp.copy_(hp0)
        us0 = timed_graph(lambda: [skinny(w, tl, sc) for w in ws], len(ws))
        gflop = 2.0 * M * Nn * K / 1e9
        print(f"rows {rows:3d} {kind:5s} skinny {str(tl):12s} passes {sc[0]}  {us0:7.2f} us  {gflop / us0 * 1e3:6.0f} TFLOP/s "
              f"{Nn * K * 2 / us0 / 1e3:6.0f} GB/s", flush=True)
        out.append(dict(rows=rows, kind=kind, impl="skinny", tile=tl, us=round(us0, 3)))
        for s in shapes:
            if kind == "w13" and s[3] % 2:
                continue
            reset()
            rc = tile(ws[0], s)
            if rc == L.ERR_UNSUPPORTED:
                continue
            if rc != 0:
                print(f"rows {rows:3d} {kind:5s} tile {s}: rc {rc}", flush=True)
                continue
            torch.cuda.synchronize()
            got = outs()
            frac, ulps = zip(*[ulp_diff(a, b) for a, b in zip(got, ref)])
            if kind in ("wo", "w2"):
                e.hp.copy_(hp0)
            us = timed_graph(lambda: [tile(w, s) for w in ws], len(ws))
            ok = max(ulps[:1 if kind in ("wo", "w2") else len(ulps)]) <= 2.0 and max(frac) < 0.05
            print(f"rows {rows:3d} {kind:5s} tile   {str(s):20s} {us:7.2f} us  {gflop / us * 1e3:6.0f} TFLOP/s {Nn * K * 2 / us / 1e3:6.0f} GB/s"
                  f"  diff frac {max(frac):.4f} max {max(ulps):.2f} ulp {'ok' if ok else 'MISMATCH'}", flush=True)
            out.append(dict(rows=rows, kind=kind, impl="tile", shape=s, us=round(us, 3), frac=max(frac), ulps=max(ulps), ok=ok))


def main():
    args = sys.argv[1:]
    name = "GPT-L"
    if args and not args[0].isdigit():
        name = args.pop(0)
    rows_list = [int(a) for a in args] or [256, 128]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    m = GPT_models[name](vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    for blk in m.layers:   # non-trivial norm weights so that the fused norm's second rounding is exercised
        torch.nn.init.normal_(blk.attention_norm.weight, 1.0, 0.2)
        torch.nn.init.normal_(blk.ffn_norm.weight, 1.0, 0.2)
    torch.nn.init.normal_(m.norm.weight, 1.0, 0.2)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    out = []
    for rows in rows_list:
        sweep(m, rows, out)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(model=name, results=out), open(f"gpurun_out/gemm_tile_sweep_{name}.json", "w"))


if __name__ == "__main__":
    main()
