"""Decode-chain GEMM timing sweep (development aid, round 3): for 64 / 128 / 256 chain rows, every GEMM kind of the
GPT-L decode step x a list of workgroup shapes (mt, nt, kw), timed as a captured chain over all 24 layers' weights
(nothing cache-resident) -> us per launch.  Prints one line per configuration and, per (rows, kind), the best shape
as an LGEN_TILES string.
    python tools/gemm_sweep.py [rows ...]            (default 64 128 256)
    LGEN_SWEEP_IMPL=xl python tools/gemm_sweep.py    (also times the round-3 kernels where the library has them)
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


def sweep(m, rows, out):
    N = 576
    m._engine = None
    torch.cuda.empty_cache()
    m.setup_caches(rows, 1 + N, torch.bfloat16)
    e = m._engine
    lib, dt, M, mts = e.lib, e.dt, e.B2, e.MTs
    d, F, H, hd, hdp, S8, V = e.d, e.F, e.H, e.hd, e.hdp, e.S8, e.V
    e.hp.normal_(0, 1)
    e.ap.normal_(0, 1)
    e.gp.normal_(0, 1)
    e.ssq.fill_(16.0)
    e.ssq_parts = d // 16
    e.state.zero_()
    nl = len(e.layers)

    e.use_tile = False   # this tool sweeps the SKINNY kernels (the big-M tile family: tools/gemm_tile_sweep.py)

    def qkv(tl, sc=(1, 0)):
        for w in e.layers:
            L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(e.hp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                           L.ptr(e.freqs_cis), L.ptr(e.state), M, mts, d, H, hd, hdp, S8, e.kvs, dt, tl[0], tl[1], tl[2],
                                           L.ptr(w["an"]), L.ptr(e.ssq), e.ssq_parts, e.eps, sc[0], L.stream()), "qkv")

    kinds = {
        "qkv": (qkv, nl, 3 * d * d * 2),
        "wo": (lambda tl, sc=None: [e.gemm(w["wo"], e.ap, e.hp, M, mts, d, d, L.EPI_RES, tl, ssq_out=e.ssq) for w in e.layers], nl, d * d * 2),
        "w13": (lambda tl, sc=None: [e.gemm(w["w13"], e.hp, e.gp, M, mts, 2 * F, d, L.EPI_SWIGLU, tl, norm_w=w["fn"], sched=sc) for w in e.layers], nl, 2 * F * d * 2),
        "w2": (lambda tl, sc=None: [e.gemm(w["w2"], e.gp, e.hp, M, mts, d, F, L.EPI_RES, tl, ssq_out=e.ssq) for w in e.layers], nl, F * d * 2),
        "head": (lambda tl, sc=None: [e.gemm(e.out_w, e.hp, e.logits, M, mts, V, d, L.EPI_ROWS, tl, norm_w=e.norm_w, sched=sc) for _ in range(4)], 4, V * d * 2),
    }
    scheds = [(1, 0), (2, 0), (3, 0), (4, 0), (6, 0), (8, 0)]   # (passes, unused): ABI v7 dropped the double-buffered mode
    cands = {
        # kw = 4 shapes of the fused GEMMs fall through to the ring kernel's RMSNorm prologue (the register-resident form needs
        # 8 waves x 3..6 chunks): the wide-model sweep (profiles/r03_wide_sweep.log) says big tiles x 4 waves are worth a look
        "qkv": [(1, 4, 8), (2, 4, 8), (2, 2, 8), (4, 2, 8), (4, 4, 8), (4, 2, 4), (2, 4, 4)],
        "w13": [(1, 4, 8), (2, 4, 8), (2, 2, 8), (4, 2, 8), (4, 4, 8), (4, 4, 4), (4, 2, 4), (2, 4, 4)],
        "head": [(1, 4, 8), (2, 4, 8), (4, 2, 8), (4, 4, 8), (4, 4, 4), (4, 2, 4)],
        "wo": [(1, 1, 8), (2, 1, 8), (4, 1, 8), (2, 2, 8), (2, 1, 4), (2, 2, 4), (4, 2, 4)],
        "w2": [(1, 1, 16), (2, 1, 16), (2, 1, 8), (4, 1, 8), (2, 2, 8), (1, 2, 16), (2, 2, 4), (4, 1, 4), (4, 2, 4)],
    }
    best = {}
    for kind, (fn, launches, nbytes) in kinds.items():
        default = e._tiles({"head": "head"}.get(kind, kind), {"qkv": 3 * d, "wo": d, "w13": 2 * F, "w2": d, "head": V}[kind],
                           F if kind == "w2" else d)
        for tl in [default] + [c for c in cands[kind] if c != default]:
            if tl[0] > mts or mts % tl[0]:
                continue
            for sc in (scheds if kind in ("qkv", "w13", "head") else [(1, 0)]):
                units = ({"qkv": 3 * d, "w13": 2 * F, "head": V}.get(kind, d) // 16 // tl[1]) * (mts // tl[0])
                if sc[0] > 1 and units // sc[0] < 64:
                    continue  # fewer than 64 workgroups: not worth a measurement
                try:
                    us = timed_graph(lambda: fn(tl, sc), launches)
                except Exception as ex:  # noqa: BLE001 -- an uninstantiated / unsupported shape
                    torch.cuda.synchronize()
                    print(f"rows {rows:3d} {kind:5s} {tl}: unsupported ({str(ex)[:60]})", flush=True)
                    break
                tag = " (default tile)" if tl == default else ""
                print(f"rows {rows:3d} {kind:5s} {str(tl):12s} passes {sc[0]} db {sc[1]}  {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s{tag}", flush=True)
                out.append(dict(rows=rows, kind=kind, tile=tl, sched=sc, us=round(us, 3), default=tl == default))
                if kind not in best or us < best[kind][2]:
                    best[kind] = (tl, sc, us)
    spec = ";".join(f"{k}={t[0]},{t[1]},{t[2]}" for k, (t, _, _) in best.items())
    pspec = ";".join(f"{k}={sc[0]},{sc[1]}" for k, (_, sc, _) in best.items() if k in ("qkv", "w13", "head"))
    tot = sum(us * (1 if k == "head" else nl) for k, (_, _, us) in best.items())
    print(f"rows {rows}: best LGEN_TILES='{spec}' LGEN_PASSES='{pspec}'  -> {tot:.0f} us of GEMMs per decode step", flush=True)
    return dict(tiles=spec, passes=pspec)


def main():
    rows_list = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    m = GPT_models["GPT-L"](vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    out, specs = [], {}
    for rows in rows_list:
        specs[rows] = sweep(m, rows, out)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(results=out, best=specs), open("gpurun_out/gemm_sweep.json", "w"))


if __name__ == "__main__":
    main()
