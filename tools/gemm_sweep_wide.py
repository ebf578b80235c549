"""Decode-chain GEMM timing for WIDE models (development aid, round 3): GPT-3B (d 3200, F 8704: the fused-norm kernels do not
apply, every GEMM is the plain ring kernel behind a stand-alone RMSNorm) at 256 chain rows, every GEMM kind x a list of
workgroup shapes (mt, nt, kw), each timed as a captured chain over all layers' weights (nothing cache-resident), with the
steady-state kernel (gemm_skinny.hip: gemm_steady_kernel) on and off (LGEN_GEMM_STEADY=0).
    python tools/gemm_sweep_wide.py [GPT-3B|GPT-XXL] [rows]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402
from tools.gemm_sweep import timed_graph  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "GPT-3B"
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    m = GPT_models[name](vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    N = 576
    m.setup_caches(rows, 1 + N, torch.bfloat16)
    e = m._engine
    lib, dt, M, mts = e.lib, e.dt, e.B2, e.MTs
    d, F, H, hd, hdp, S8, V = e.d, e.F, e.H, e.hd, e.hdp, e.S8, e.V
    for t in (e.hp, e.ap, e.gp, e.xnp):
        t.normal_(0, 1)
    e.state.zero_()
    nl = len(e.layers)
    fused = e.fuse_norm
    print(f"{name}: d {d} F {F} rows {M} (MTs {mts}), {nl} layers, fused-norm kernels {'on' if fused else 'off'}", flush=True)

    def qkv(tl):
        for w in e.layers:
            L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(e.xnp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                           L.ptr(e.freqs_cis), L.ptr(e.state), M, mts, d, H, hd, hdp, S8, e.kvs, dt, tl[0], tl[1], tl[2],
                                           0, 0, 0, e.eps, L.stream()), "qkv")

    kinds = {
        "qkv": (qkv, nl, 3 * d, d),
        "wo": (lambda tl: [e.gemm(w["wo"], e.ap, e.hp, M, mts, d, d, L.EPI_RES, tl) for w in e.layers], nl, d, d),
        "w13": (lambda tl: [e.gemm(w["w13"], e.xnp, e.gp, M, mts, 2 * F, d, L.EPI_SWIGLU, tl) for w in e.layers], nl, 2 * F, d),
        "w2": (lambda tl: [e.gemm(w["w2"], e.gp, e.hp, M, mts, d, F, L.EPI_RES, tl) for w in e.layers], nl, d, F),
        "head": (lambda tl: [e.gemm(e.out_w, e.xnp, e.logits, M, mts, V, d, L.EPI_ROWS, tl) for _ in range(4)], 4, V, d),
    }
    if fused:  # GPT-XXL: only the plain GEMMs are of interest here
        kinds = {k: v for k, v in kinds.items() if k in ("wo", "w2")}
    best = {}
    for kind, (fn, launches, n_out, k_in) in kinds.items():
        kch = k_in // 32
        default = e._tiles(kind, n_out, k_in)
        kws = sorted({k for k in (4, 5, 8, 10, 16) if kch % k == 0 and kch // k >= 6} | {default[2]})
        shapes = [(4, 1), (4, 2), (4, 4), (8, 1), (8, 2), (2, 2), (2, 4)]
        cands = [default] + [(a, b, k) for a, b in shapes for k in kws if (a, b, k) != default]
        for tl in cands:
            if tl[0] > mts or mts % tl[0] or (n_out // 16) % tl[1] or (kind == "w13" and tl[1] & 1):
                continue
            res = {}
            for steady in ("1", "0"):
                os.environ["LGEN_GEMM_STEADY"] = steady
                try:
                    res[steady] = timed_graph(lambda: fn(tl), launches)
                except Exception as ex:  # noqa: BLE001 -- an unsupported shape
                    torch.cuda.synchronize()
                    res[steady] = None
                    print(f"{kind:5s} {tl}: unsupported ({str(ex)[:50]})", flush=True)
                    break
            if res.get("1") is None:
                continue
            flops = 2.0 * M * n_out * k_in
            tag = " (default)" if tl == default else ""
            off = f"{res['0']:7.2f}" if res.get("0") else "   n/a"
            print(f"{kind:5s} {str(tl):12s} steady {res['1']:7.2f} us  generic {off} us   {flops / res['1'] / 1e6:6.0f} TFLOP/s  "
                  f"{n_out * k_in * 2 / res['1'] / 1e3:6.0f} GB/s{tag}", flush=True)
            if kind not in best or res["1"] < best[kind][1]:
                best[kind] = (tl, res["1"])
    os.environ.pop("LGEN_GEMM_STEADY", None)
    spec = ";".join(f"{k}={t[0]},{t[1]},{t[2]}" for k, (t, _) in best.items())
    tot = sum(us * (1 if k == "head" else nl) for k, (_, us) in best.items())
    print(f"{name} rows {rows}: best LGEN_TILES='{spec}' -> {tot:.0f} us of these GEMMs per decode step", flush=True)


if __name__ == "__main__":
    main()
