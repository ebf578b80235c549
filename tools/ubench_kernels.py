"""Per-kernel cost inside a dependent chain (development aid): graphs that contain ONLY one kind of
decode kernel (all 24 layers' weights in turn, so nothing is cache-resident), timed per launch."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import GPT_models, generate  # noqa: E402
from llamagen_amd import _lib as L  # noqa: E402


def timed_graph(fn, reps=10):
    fn()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main(name="GPT-L", B=32, img=384):
    dev = torch.device("cuda:0")
    lat = img // 16
    N = lat * lat
    torch.manual_seed(0)
    m = GPT_models[name](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    m = m.to(device=dev, dtype=torch.bfloat16).eval()
    c = torch.randint(0, 1000, (B,), device=dev)
    kw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    generate(m, c, N, **kw)
    torch.cuda.synchronize()
    e = m._engine
    lib, dt, M, mts = e.lib, e.dt, e.B2, e.MTs
    d, F, H, hd, hdp, S8, V = e.d, e.F, e.H, e.hd, e.hdp, e.S8, e.V
    e.k_cache.normal_(0, 1)
    e.v_cache.normal_(0, 1)
    st = lambda: L.stream()
    nl = len(e.layers)
    os.system("rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk\\|fclk' | head -8")

    def report(tag, fn, launches, nbytes=None):
        t = timed_graph(fn)
        per = t / launches
        extra = f"  {nbytes / per / 1e3:8.1f} GB/s" if nbytes else ""
        print(f"{tag:52s} {per:7.2f} us/launch{extra}", flush=True)

    # trivial kernels
    report("advance_state (1 thread)", lambda: [lib.lgen_advance_state(L.ptr(e.state), st()) for _ in range(100)], 100)
    e.state.zero_()
    report("embed_pack (+ssq)", lambda: [e._embed(e.tok_emb, e.cur_tok) for _ in range(50)], 50)
    e.ssq_parts = d // 16
    for kind, tiles_list in {
        "qkv": [(1, 4, 8)],
        "wo": [(1, 1, 8)],
        "w13": [(2, 4, 8)],
        "w2": [(1, 1, 16)],
    }.items():
        for tl in tiles_list:
            for norm, hot in (((True, False), (True, True)) if kind in ("qkv", "w13") else ((False, False), (False, True))):
                def fn():
                    for w in ([e.layers[0]] * nl if hot else e.layers):
                        if kind == "qkv":
                            L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(e.hp), L.ptr(e.qbuf), L.ptr(e.k_cache[0]), L.ptr(e.v_cache[0]),
                                                           L.ptr(e.freqs_cis), L.ptr(e.state), M, mts, d, H, hd, hdp, S8, e.kvs, dt, tl[0], tl[1], tl[2],
                                                           L.ptr(w["an"]) if norm else 0, L.ptr(e.ssq), e.ssq_parts, e.eps, st()), "qkv")
                        elif kind == "wo":
                            e.gemm(w["wo"], e.ap, e.hp, M, mts, d, d, L.EPI_RES, tl, ssq_out=e.ssq)
                        elif kind == "w13":
                            e.gemm(w["w13"], e.hp, e.gp, M, mts, 2 * F, d, L.EPI_SWIGLU, tl, norm_w=w["fn"] if norm else None)
                        else:
                            e.gemm(w["w2"], e.gp, e.hp, M, mts, d, F, L.EPI_RES, tl, ssq_out=e.ssq)
                nb = {"qkv": 3 * d * d, "wo": d * d, "w13": 2 * F * d, "w2": F * d}[kind] * 2
                try:
                    report(f"{kind} {tl} norm={norm} weights {'MALL-hot (same layer)' if hot else 'cold (24 layers)'}", fn, nl, nb)
                except Exception as ex:
                    print(kind, tl, norm, "->", ex)
    for tl in [(4, 2, 8), (4, 4, 8), (4, 4, 4)]:
        report(f"head {tl} norm=True", lambda: [e.gemm(e.out_w, e.hp, e.logits, M, mts, V, d, L.EPI_ROWS, tl, norm_w=e.norm_w) for _ in range(8)], 8, V * d * 2)
    report("rmsnorm standalone", lambda: [L.check(lib.lgen_rmsnorm(L.ptr(e.hp), L.ptr(e.norm_w), L.ptr(e.xnp), mts, d, e.eps, dt, st()), "n") for _ in range(48)], 48)
    for variant in (2, 4, 5):
        lib.lgen_set_attn_variant(variant)
        for pos in (0, 64, 144, 288, 575):
            e.state.copy_(torch.tensor([pos, pos], dtype=torch.int32, device=dev))
            def fa():
                for i in range(nl):
                    L.check(lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap), L.ptr(e.state), 0, 0,
                                                 M, mts, H, hd, hdp, S8, e.kvs, dt, st()), "attn")
            report(f"attn variant {variant} pos {pos}", fa, nl, (pos + 1) * 2 * H * hd * 2 * M)
    lib.lgen_set_attn_variant(2)
    sp = dict(use_cfg=True, cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    e.noise = torch.empty(8, B, V, device=dev).exponential_(1.0)
    e.state.zero_()
    report("sampler (cfg, top-k 2000)", lambda: [e._sample(B, sp) for _ in range(20)], 20)
    x = torch.zeros(64, device=dev)
    report("torch add_ (64 floats)", lambda: [x.add_(1.0) for _ in range(100)], 100)
    os.system("rocm-smi --showclocks 2>/dev/null | grep -i 'sclk\\|mclk\\|fclk' | head -8")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "GPT-L", int(a[1]) if len(a) > 1 else 32, int(a[2]) if len(a) > 2 else 384)
