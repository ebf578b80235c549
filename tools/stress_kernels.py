"""Determinism / race stress of every kernel family of liblgen_hip.so (round 6).

Each case launches ONE library entry point `--iters` times on fixed inputs; before every launch every output buffer is poisoned
(0xFF.. = NaN for every float type), after it the outputs are compared BITWISE with the first launch's on the device (no host
sync in the loop: a per-iteration mismatch count and the union of the differing positions are accumulated by device ops), and the
inputs are compared bitwise with their copies at the end (a kernel that writes outside its outputs shows here).

    python tools/stress_kernels.py [--iters 5000] [--only substr] [--fresh] [--hammer] [--harness N]

--fresh    : outputs are allocated anew for every launch (the caching allocator hands back recently used blocks) instead of reused
--hammer   : a second stream streams HBM (1 GiB device copies) while the kernels run
--harness N: additionally replay the HARNESS of tests/test_gpu_gpt.py::test_qkv_rope_append_and_attention[192-16-64-8-40-False]
             N times (fresh host->device copies, pack_act / pack_weight, launch, alternating bf16 / fp16 like the suite's neighbours)
Run it also under AMD_SERIALIZE_KERNEL=3 (and HIP_LAUNCH_BLOCKING=1): a failure that disappears there is a cross-kernel hazard.
Exit code 1 when any case saw a mismatch.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from llamagen_amd import _lib as L  # noqa: E402
from llamagen_amd.engine import (TILE_SCHEDULES, pack_act, pack_weight, precompute_freqs_cis_2d)  # noqa: E402

DEV = torch.device("cuda:0")
DTS = {"bf16": (torch.bfloat16, L.BF16), "fp16": (torch.float16, L.F16), "fp32": (torch.float32, L.F32)}


def rnd(shape, dt, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt)


def bits(t):
    """integer view of a tensor for bitwise comparison"""
    t = t.contiguous() if not t.is_contiguous() else t
    return t.view({1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])


def poison(t):
    bits(t).fill_(-1) if t.element_size() > 1 else t.fill_(255)


class Case:
    """name; launch(outs) -> rc; make_outs() -> list of fresh output tensors (poisoned by the loop); restore(outs): in/out operands"""

    def __init__(self, name, launch, make_outs, ins, restore=None):
        self.name, self.launch, self.make_outs, self.ins, self.restore = name, launch, make_outs, ins, restore


def run_case(c, iters, fresh, hammer):
    outs = c.make_outs()
    in_copies = [t.clone() for t in c.ins]
    for o in outs:
        poison(o)
    if c.restore:
        c.restore(outs)
    rc = c.launch(outs)
    if rc == L.ERR_UNSUPPORTED:
        return None
    assert rc == 0, (c.name, rc)
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    nbad = torch.zeros(iters, dtype=torch.int64, device=DEV)
    union = [torch.zeros(o.shape, dtype=torch.bool, device=DEV) for o in outs]
    t0 = time.time()
    for i in range(iters):
        if hammer is not None:
            hammer.tick()
        if fresh:
            outs = c.make_outs()
        for o in outs:
            poison(o)
        if c.restore:
            c.restore(outs)
        rc = c.launch(outs)
        assert rc == 0, (c.name, rc)
        cnt = None
        for o, r, u in zip(outs, ref, union):
            m = bits(o) != bits(r)
            u |= m.view(u.shape) if m.shape == u.shape else m.reshape(u.shape)
            s = m.sum()
            cnt = s if cnt is None else cnt + s
        nbad[i] = cnt
    torch.cuda.synchronize()
    dt = time.time() - t0
    ins_ok = all(torch.equal(bits(a), bits(b)) for a, b in zip(c.ins, in_copies))
    nb = nbad.cpu()
    bad_iters = torch.nonzero(nb).flatten().tolist()
    rec = dict(case=c.name, iters=iters, bad_launches=len(bad_iters), inputs_intact=ins_ok, seconds=round(dt, 2))
    if bad_iters:
        rec["first_bad_iterations"] = bad_iters[:10]
        rec["elements_per_bad_launch"] = nb[bad_iters[:10]].tolist()
        rec["union_positions"] = [torch.nonzero(u)[:24].tolist() for u in union]
    return rec


class Hammer:
    """HBM traffic on a second stream: 1 GiB device-to-device copies, at most 4 queued"""

    def __init__(self):
        self.s = torch.cuda.Stream()
        self.a = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
        self.b = torch.empty(1 << 28, dtype=torch.float32, device=DEV)
        self.ev = []

    def tick(self):
        while len(self.ev) >= 4:
            self.ev.pop(0).synchronize()
        with torch.cuda.stream(self.s):
            self.b.copy_(self.a, non_blocking=True)
            e = torch.cuda.Event()
            e.record(self.s)
        self.ev.append(e)


# ------------------------------------------------------------------------------------------------------------------ cases
def qkv_cases():
    lib = L.lib()
    out = []
    for dname, M, d, H, hd, tiles, norm in [
            ("fp16", 192, 1024, 16, 64, (4, 1, 4), False), ("bf16", 192, 1024, 16, 64, (4, 1, 4), False),
            ("fp32", 192, 1024, 16, 64, (4, 1, 4), False), ("bf16", 64, 1024, 16, 64, (1, 4, 8), True),
            ("bf16", 128, 1024, 16, 64, (2, 4, 8), True), ("bf16", 128, 3200, 32, 100, (4, 1, 8), False),
            ("fp16", 33, 1024, 16, 64, (4, 1, 4), False), ("bf16", 64, 1536, 24, 64, (1, 4, 8), True)]:
        dt, code = DTS[dname]
        hdp = 64 if hd <= 64 else 128
        grid, pos = 24, 300
        S8 = (1 + grid * grid + 7) // 8 * 8
        mts = (M + 15) // 16
        mts = {3: 4}.get(mts, mts)
        x, w = rnd((M, d), dt, 9), rnd((3 * d, d), dt, 10, 0.05)
        nw = (1 + 0.1 * rnd((d,), torch.float32, 43)).to(dt).to(DEV)
        xp, wp = pack_act(x.to(DEV), mts), pack_weight(w.to(DEV))
        fr = precompute_freqs_cis_2d(grid, hd, 10000.0, 1).to(DEV)
        state = torch.tensor([pos, 0], dtype=torch.int32, device=DEV)
        ssq = torch.zeros(mts * 16, L.SSQ_STRIDE, device=DEV)
        if norm:
            L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, code, L.stream()), "ssq")
        mt, nt, kw = tiles
        mt = min(mt, mts)

        def make(M=M, H=H, S8=S8, hdp=hdp, mts=mts, dt=dt):
            # (a small cache: 8 slots around `pos` would change the addressing; keep the full slab but only 2 rows deep where possible)
            return [torch.empty(mts * 16, H, hdp, dtype=dt, device=DEV), torch.empty(M, H, S8, hdp, dtype=dt, device=DEV),
                    torch.empty(M, H, S8, hdp, dtype=dt, device=DEV)]

        def launch(o, wp=wp, xp=xp, fr=fr, state=state, M=M, mts=mts, d=d, H=H, hd=hd, hdp=hdp, S8=S8, code=code, mt=mt, nt=nt, kw=kw,
                   nw=nw, ssq=ssq, norm=norm):
            return lib.lgen_gemm_qkv_rope(L.ptr(wp), L.ptr(xp), L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.ptr(fr), L.ptr(state), M, mts,
                                          d, H, hd, hdp, S8, 0, code, mt, nt, kw, L.ptr(nw) if norm else 0, L.ptr(ssq) if norm else 0,
                                          d // 16 if norm else 0, 1e-5, 1, L.stream())

        out.append(Case(f"skinny qkv_rope {dname} M{M} d{d} hd{hd} tiles{tiles} norm{int(norm)}", launch, make, [xp, wp, fr, state, ssq, nw]))
    return out


def gemm_cases():
    lib = L.lib()
    out = []
    shapes = [(64, 3072, 1024, (4, 1, 8)), (64, 1024, 2816, (4, 1, 8)), (64, 2048, 1024, (4, 4, 2)), (33, 512, 800, (4, 2, 3)),
              (128, 512, 1024, (8, 2, 4)), (128, 512, 1024, (8, 1, 8)), (256, 256, 512, (8, 1, 2)), (5, 1024, 512, (1, 2, 4)),
              (192, 1024, 1024, (4, 1, 4)), (256, 1024, 3200, (8, 2, 5)), (128, 512, 8704, (4, 2, 8))]
    for dname in ("bf16", "fp16", "fp32"):
        dt, code = DTS[dname]
        kc = 16 if dname == "fp32" else 32
        for M, N, K, tiles in shapes:
            if K % kc or (dname != "bf16" and K > 3000):
                continue
            mts = (M + 15) // 16
            mts = {3: 4}.get(mts, mts)
            if mts > 4:
                mts = (mts + 7) // 8 * 8
            mt, nt, kw = tiles
            mt = min(mt, mts)
            x, w = rnd((M, K), dt, 3), rnd((N, K), dt, 4, 0.05)
            xp, wp = pack_act(x.to(DEV), mts), pack_weight(w.to(DEV))
            h0 = pack_act(rnd((M, N), dt, 5).to(DEV), mts) if N % kc == 0 else None
            for epi, ename in ((L.EPI_ROWS, "rows"), (L.EPI_PACKED, "packed"), (L.EPI_RES, "res+ssq"), (L.EPI_GELU, "gelu"),
                               (L.EPI_SWIGLU, "swiglu")):
                if epi != L.EPI_ROWS and N % kc:
                    continue
                if epi == L.EPI_SWIGLU and (nt % 2 or (N // 2) % kc):
                    continue

                def make(epi=epi, mts=mts, N=N, kc=kc, dt=dt):
                    if epi == L.EPI_ROWS:
                        return [torch.empty(mts * 16, N, dtype=dt, device=DEV)]
                    if epi == L.EPI_SWIGLU:
                        return [torch.empty(N // 2 // kc, mts, 64, kc // 4, dtype=dt, device=DEV)]
                    o = [torch.empty(N // kc, mts, 64, kc // 4, dtype=dt, device=DEV)]
                    if epi == L.EPI_RES:
                        o.append(torch.empty(mts * 16, L.SSQ_STRIDE, device=DEV))
                    return o

                def restore(o, epi=epi, h0=h0):
                    if epi == L.EPI_RES:
                        o[0].copy_(h0.view(o[0].shape))

                def launch(o, wp=wp, xp=xp, M=M, mts=mts, N=N, K=K, epi=epi, code=code, mt=mt, nt=nt, kw=kw):
                    return lib.lgen_gemm(L.ptr(wp), L.ptr(xp), L.ptr(o[0]), M, mts, N, K, epi, code, mt, nt, kw, 0, 0, 0, 0.0,
                                         L.ptr(o[1]) if epi == L.EPI_RES else 0, 1, L.stream())

                out.append(Case(f"skinny gemm {ename} {dname} M{M} N{N} K{K} tiles{tiles}", launch, make, [xp, wp] + ([h0] if h0 is not None else []),
                                restore))
    # fused RMSNorm forms (gemm_normpre.hip and the NORM = true ring kernel)
    dt, code = DTS["bf16"]
    for M, d, N, epi, tiles, passes in [(64, 1024, 5632, L.EPI_SWIGLU, (1, 4, 8), 1), (64, 1024, 5632, L.EPI_SWIGLU, (1, 4, 8), 3),
                                        (128, 1024, 16384, L.EPI_ROWS, (2, 4, 8), 2), (64, 1280, 7168, L.EPI_SWIGLU, (1, 4, 8), 1),
                                        (33, 800, 256, L.EPI_ROWS, (4, 1, 5), 1), (128, 1024, 512, L.EPI_SWIGLU, (4, 2, 4), 1)]:
        mts = (M + 15) // 16
        mts = {3: 4}.get(mts, mts)
        x, w = rnd((M, d), dt, 31, 1.3), rnd((N, d), dt, 33, 0.03)
        nw = (1 + 0.1 * rnd((d,), torch.float32, 34)).to(dt).to(DEV)
        xp, wp = pack_act(x.to(DEV), mts), pack_weight(w.to(DEV))
        ssq = torch.zeros(mts * 16, L.SSQ_STRIDE, device=DEV)
        L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, code, L.stream()), "ssq")
        mt, nt, kw = tiles
        mt = min(mt, mts)

        def make(epi=epi, mts=mts, N=N, dt=dt):
            return [torch.empty(mts * 16, N, dtype=dt, device=DEV)] if epi == L.EPI_ROWS else [torch.empty(N // 2 // 32, mts, 64, 8, dtype=dt, device=DEV)]

        def launch(o, wp=wp, xp=xp, M=M, mts=mts, N=N, d=d, epi=epi, mt=mt, nt=nt, kw=kw, nw=nw, ssq=ssq, passes=passes, code=code):
            return lib.lgen_gemm(L.ptr(wp), L.ptr(xp), L.ptr(o[0]), M, mts, N, d, epi, code, mt, nt, kw, L.ptr(nw), L.ptr(ssq), d // 16, 1e-5,
                                 0, passes, L.stream())

        out.append(Case(f"skinny fused-norm gemm epi{epi} bf16 M{M} d{d} N{N} tiles{tiles} passes{passes}", launch, make, [xp, wp, nw, ssq]))
    return out


def tile_cases():
    lib = L.lib()
    out = []
    dt, code = DTS["bf16"]
    d, H, hd, F, V = 1024, 16, 64, 2816, 16384
    grid, pos = 24, 300
    S8 = (1 + grid * grid + 7) // 8 * 8
    for M in (256, 640):
        mts = M // 16
        t = TILE_SCHEDULES[16 if M == 256 else 40]
        x = rnd((M, d), dt, 61, 1.3)
        nw = (1 + 0.1 * rnd((d,), torch.float32, 62)).to(dt).to(DEV)
        wq, w13w, wh = rnd((3 * d, d), dt, 63, 0.03), rnd((2 * F, d), dt, 64, 0.03), rnd((V, d), dt, 66, 0.03)
        wo, w2 = rnd((d, d), dt, 67, 0.03), rnd((d, F), dt, 68, 0.02)
        xp = pack_act(x.to(DEV), mts)
        ap, gp_in, hp0 = pack_act(rnd((M, d), dt, 69).to(DEV), mts), pack_act(rnd((M, F), dt, 70, 0.7).to(DEV), mts), pack_act(rnd((M, d), dt, 71, 1.5).to(DEV), mts)
        wqp, w13p, whp, wop, w2p = (pack_weight(w_.to(DEV)) for w_ in (wq, w13w, wh, wo, w2))
        fr = precompute_freqs_cis_2d(grid, hd, 10000.0, 1).to(DEV)
        state = torch.tensor([pos, 0], dtype=torch.int32, device=DEV)
        ssq = torch.zeros(mts * 16, L.SSQ_STRIDE, device=DEV)
        L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, code, L.stream()), "ssq")

        def mk_qkv(M=M, mts=mts):
            return [torch.empty(mts * 16, H, 64, dtype=dt, device=DEV), torch.empty(M, H, S8, 64, dtype=dt, device=DEV),
                    torch.empty(M, H, S8, 64, dtype=dt, device=DEV)]

        def l_qkv(o, s=t["qkv"], wqp=wqp, xp=xp, fr=fr, state=state, M=M, mts=mts, nw=nw, ssq=ssq):
            return lib.lgen_gemm_qkv_rope_tile(L.ptr(wqp), L.ptr(xp), L.ptr(o[0]), L.ptr(o[1]), L.ptr(o[2]), L.ptr(fr), L.ptr(state), M, mts, d,
                                               H, hd, 64, S8, 0, code, *s, L.ptr(nw), L.ptr(ssq), d // 16, 1e-5, L.stream())

        out.append(Case(f"tile qkv_rope M{M} {t['qkv']}", l_qkv, mk_qkv, [wqp, xp, fr, state, nw, ssq]))

        def l_w13(o, s=t["w13"], w13p=w13p, xp=xp, M=M, mts=mts, nw=nw, ssq=ssq):
            return lib.lgen_gemm_tile(L.ptr(w13p), L.ptr(xp), L.ptr(o[0]), M, mts, 2 * F, d, L.EPI_SWIGLU, code, *s, L.ptr(nw), L.ptr(ssq),
                                      d // 16, 1e-5, 0, L.stream())

        out.append(Case(f"tile w13 swiglu M{M} {t['w13']}", l_w13, lambda mts=mts: [torch.empty(F // 32, mts, 64, 8, dtype=dt, device=DEV)],
                        [w13p, xp, nw, ssq]))

        def l_head(o, s=t["head"], whp=whp, xp=xp, M=M, mts=mts, nw=nw, ssq=ssq):
            return lib.lgen_gemm_tile(L.ptr(whp), L.ptr(xp), L.ptr(o[0]), M, mts, V, d, L.EPI_ROWS, code, *s, L.ptr(nw), L.ptr(ssq), d // 16,
                                      1e-5, 0, L.stream())

        out.append(Case(f"tile lm_head M{M} {t['head']}", l_head, lambda mts=mts: [torch.empty(mts * 16, V, dtype=dt, device=DEV)], [whp, xp, nw, ssq]))
        for name, wp_, xin, K, s in (("wo", wop, ap, d, t["wo"]), ("w2", w2p, gp_in, F, t["w2"])):
            def l_res(o, wp_=wp_, xin=xin, K=K, s=s, M=M, mts=mts):
                return lib.lgen_gemm_tile(L.ptr(wp_), L.ptr(xin), L.ptr(o[0]), M, mts, d, K, L.EPI_RES, code, *s, 0, 0, 0, 0.0, L.ptr(o[1]),
                                          L.stream())

            out.append(Case(f"tile {name} res+ssq M{M} {s}", l_res,
                            lambda mts=mts: [torch.empty(d // 32, mts, 64, 8, dtype=dt, device=DEV), torch.empty(mts * 16, L.SSQ_STRIDE, device=DEV)],
                            [wp_, xin, hp0], lambda o, hp0=hp0: o[0].copy_(hp0.view(o[0].shape))))
    return out


def attn_cases():
    lib = L.lib()
    out = []
    for dname, B2, H, hd, pos, variants in [("bf16", 64, 16, 64, 300, (-1, 0, 6)), ("bf16", 640, 16, 64, 300, (-1, 8, 11)),
                                           ("bf16", 640, 16, 64, 575, (-1,)), ("fp16", 192, 16, 64, 40, (-1, 8, 13)),
                                           ("bf16", 128, 32, 100, 300, (-1,)), ("fp32", 33, 16, 64, 300, (-1, 2))]:
        dt, code = DTS[dname]
        hdp = 64 if hd <= 64 else 128
        S8 = 584
        mts = (B2 + 15) // 16
        mts = {3: 4}.get(mts, mts)
        kcd = 16 if dname == "fp32" else 32
        d = H * hd
        if d % kcd:
            continue
        epl = kcd // 4
        kvs = (hd + epl - 1) // epl * epl
        n = B2 * H * S8 * kvs
        g = torch.Generator(device=DEV).manual_seed(7)
        kflat = torch.zeros(n + hdp, dtype=dt, device=DEV)
        vflat = torch.zeros(n + hdp, dtype=dt, device=DEV)
        kflat[:n] = torch.randn(n, generator=g, device=DEV).to(dt)
        vflat[:n] = torch.randn(n, generator=g, device=DEV).to(dt)
        q = torch.zeros(mts * 16, H, hdp, dtype=dt, device=DEV)
        q[:B2, :, :hd] = torch.randn(B2, H, hd, generator=g, device=DEV).to(dt)
        state = torch.tensor([pos, 0], dtype=torch.int32, device=DEV)
        for v in variants:
            def launch(o, q=q, kflat=kflat, vflat=vflat, state=state, B2=B2, mts=mts, H=H, hd=hd, hdp=hdp, kvs=kvs, code=code, v=v):
                return lib.lgen_attn_decode(L.ptr(q), L.ptr(kflat), L.ptr(vflat), L.ptr(o[0]), L.ptr(state), 0, 0, B2, mts, H, hd, hdp, S8, kvs,
                                            code, v, L.stream())

            out.append(Case(f"attn_decode {dname} rows{B2} H{H} hd{hd} pos{pos} variant{v}", launch,
                            lambda d=d, kcd=kcd, mts=mts, dt=dt: [torch.empty(d // kcd, mts, 64, kcd // 4, dtype=dt, device=DEV)],
                            [q, kflat, vflat, state]))
    return out


def misc_cases():
    lib = L.lib()
    out = []
    # sampler (CFG mix + top-k + softmax + argmax(p / q)) at the bench's shape
    for dname in ("bf16", "fp32"):
        dt, code = DTS[dname]
        B, V = 32, 16384
        logits = rnd((2 * B, V), dt, 21, 2.0).to(DEV)
        noise = torch.empty(4, B, V, device=DEV).exponential_(1.0)
        state = torch.tensor([5, 2], dtype=torch.int32, device=DEV)

        def launch(o, logits=logits, noise=noise, state=state, code=code):
            return lib.lgen_sample(L.ptr(logits), L.ptr(noise), B * V, L.ptr(o[0]), L.ptr(o[1]), L.ptr(state), B, V, 576, 1, 4.0, -1, 1.0, 2000,
                                   1.0, 0, code, L.stream())

        out.append(Case(f"sample {dname} B{B} V{V} cfg4 topk2000", launch,
                        lambda: [torch.empty(2 * B, dtype=torch.int32, device=DEV), torch.empty(B, 576, dtype=torch.int32, device=DEV)],
                        [logits, noise, state]))
    # rmsnorm
    dt, code = DTS["bf16"]
    xp = pack_act(rnd((64, 1024), dt, 1).to(DEV), 4)
    w = (1 + 0.1 * rnd((1024,), torch.float32, 2)).to(dt).to(DEV)
    out.append(Case("rmsnorm bf16 M64 d1024", lambda o: lib.lgen_rmsnorm(L.ptr(xp), L.ptr(w), L.ptr(o[0]), 4, 1024, 1e-5, code, L.stream()),
                    lambda: [torch.empty_like(xp)], [xp, w]))
    return out


def vq_cases():
    """Round 6: the fused 3x3 convolution with its hand-ordered fragment reads (counted lgkmcnt waits, one tap ahead: a wait that is one
    short would be an intermittent wrong result, exactly what this tool looks for) in its default and older staging variants, on a full
    and on a ragged tile row; the 1x1 form; the row-statistics grouping of GPT-3B (lgen_ssq_group4)."""
    from llamagen_amd.vq_engine import _ConvW
    lib = L.lib()
    out = []

    class Cv:
        pass
    for (B, H, W, Cin, Cout, k, gn, res) in ((2, 48, 48, 128, 128, 3, 1, 1), (2, 24, 24, 256, 256, 3, 1, 0), (1, 32, 64, 512, 128, 3, 0, 1),
                                             (2, 16, 32, 256, 128, 1, 1, 1)):
        cv = Cv()
        cv.weight = (rnd((Cout, Cin, k, k), torch.float32, 31) / (Cin * k * k) ** 0.5).to(DEV)
        cv.bias = (0.1 * rnd((Cout,), torch.float32, 32)).to(DEV)
        cw = _ConvW(cv)
        x = (rnd((B, H, W, Cin), torch.float32, 33) * 1.5 + 0.3).to(DEV)
        r = rnd((B, H, W, Cout), torch.float32, 34).to(DEV)
        coef = torch.stack([1 + 0.1 * rnd((B, Cin), torch.float32, 35), 0.1 * rnd((B, Cin), torch.float32, 36)], -1).contiguous().to(DEV)
        ntiles = (H // 8) * ((W + 15) // 16)
        for variant in ((3, 2, 1) if k == 3 else (3,)):
            def launch(o, x=x, r=r, coef=coef, cw=cw, variant=variant, B=B, H=H, W=W, Cin=Cin, Cout=Cout, k=k, gn=gn, res=res):
                lib.lgen_debug_set_conv_fused_variant(variant)
                try:
                    return lib.lgen_conv_fused(L.ptr(x), L.ptr(coef) if gn else 0, gn, L.ptr(cw.frag), L.ptr(cw.bias), L.ptr(r) if res else 0, L.ptr(o[0]),
                                               L.ptr(o[1]), B, H, W, Cin, Cout, cw.fnpad, k, 0, 0, L.stream())
                finally:
                    lib.lgen_debug_set_conv_fused_variant(3)
            # (partials of dropped tile columns are never written: they keep the poison pattern in every launch alike)
            out.append(Case(f"conv_fused {k}x{k} variant {variant} B{B} {H}x{W} {Cin}->{Cout} gn{gn} res{res}", launch,
                            lambda B=B, H=H, W=W, Cout=Cout, ntiles=ntiles, cw=cw: [torch.empty(B, H, W, Cout, device=DEV),
                                                                                    torch.empty(B, ntiles, cw.fnpad // 4, 2, device=DEV)],
                            [x, r, coef, cw.frag, cw.bias]))
    for rows, parts in ((512, 200), (256, 96), (64, 4), (128, 256)):
        ssq = rnd((rows, L.SSQ_STRIDE), torch.float32, 37).abs().to(DEV)
        out.append(Case(f"ssq_group4 rows{rows} parts{parts}", lambda o, ssq=ssq, rows=rows, parts=parts: lib.lgen_ssq_group4(L.ptr(ssq), L.ptr(o[0]), rows, parts, L.stream()),
                        lambda rows=rows: [torch.empty(rows, L.SSQ_STRIDE, device=DEV)], [ssq]))
    return out


def sample_signature_ok():
    """lgen_sample's argument order is checked against _lib.SIGNATURES before the sampler case is used"""
    return len(L.SIGNATURES["lgen_sample"]) == 18


# ---------------------------------------------------------------------------------------------------------------- harness
def harness_qkv(n, hammer):
    """The failing test's own sequence, n times: fresh CPU tensors -> .to(dev) -> pack -> launch -> compare with the first result of the
    same dtype, inputs included.  bf16 / fp16 alternate, as the suite's neighbouring cases do (same block sizes: the allocator hands
    the previous case's blocks back)."""
    lib = L.lib()
    B2, H, hd, grid, pos = 192, 16, 64, 8, 40
    d, hdp = H * hd, 64
    S8 = (1 + grid * grid + 7) // 8 * 8
    mts = 12
    first = {}
    bad = []
    for it in range(n):
        dname = ("bf16", "fp16")[it & 1]
        dt, code = DTS[dname]
        if hammer is not None:
            hammer.tick()
        x, w = rnd((B2, d), dt, 9), rnd((3 * d, d), dt, 10, 0.05)
        freqs = precompute_freqs_cis_2d(grid, hd, 10000.0, 1)
        kc_d = torch.zeros(B2, H, S8, hdp, dtype=dt, device=DEV)
        vc_d = torch.zeros(B2, H, S8, hdp, dtype=dt, device=DEV)
        kc_d[..., :hd] = rnd((B2, H, S8, hd), dt, 11).to(DEV)
        vc_d[..., :hd] = rnd((B2, H, S8, hd), dt, 12).to(DEV)
        q_d = torch.zeros(mts * 16, H, hdp, dtype=dt, device=DEV)
        state = torch.tensor([pos, 0], dtype=torch.int32, device=DEV)
        wp, xp, fr_d = pack_weight(w.to(DEV)), pack_act(x.to(DEV), mts), freqs.to(DEV)
        L.check(lib.lgen_gemm_qkv_rope(L.ptr(wp), L.ptr(xp), L.ptr(q_d), L.ptr(kc_d), L.ptr(vc_d), L.ptr(fr_d), L.ptr(state), B2, mts, d, H,
                                       hd, hdp, S8, 0, code, 4, 1, 4, 0, 0, 0, 0.0, 1, L.stream()), "qkv")
        got = dict(q=q_d.cpu(), k=kc_d[:, :, pos].cpu(), v=vc_d[:, :, pos].cpu(), xp=xp.cpu(), wp=wp.cpu())
        if dname not in first:
            first[dname] = got
            continue
        for k_, v_ in got.items():
            if not torch.equal(bits(v_), bits(first[dname][k_])):
                diff = torch.nonzero(bits(v_) != bits(first[dname][k_]))
                bad.append(dict(iteration=it, dtype=dname, tensor=k_, elements=int(diff.shape[0]), first_positions=diff[:20].tolist(),
                                max_abs=float((v_.float() - first[dname][k_].float()).abs().max())))
    return dict(case="harness qkv_rope 192 rows (test sequence, bf16/fp16 alternating)", iterations=n, bad=bad[:20], bad_count=len(bad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--only", default="")
    ap.add_argument("--fresh", action="store_true")
    ap.add_argument("--hammer", action="store_true")
    ap.add_argument("--harness", type=int, default=0)
    a = ap.parse_args()
    print("env:", {k: v for k, v in os.environ.items() if k.startswith(("AMD_SERIALIZE", "HIP_LAUNCH", "LGEN_"))}, "fresh", a.fresh,
          "hammer", a.hammer, flush=True)
    assert sample_signature_ok()
    hammer = Hammer() if a.hammer else None
    failed = 0
    for fam in (qkv_cases, gemm_cases, tile_cases, attn_cases, misc_cases, vq_cases):
        for c in fam():
            if a.only and a.only not in c.name and a.only not in fam.__name__:
                continue
            rec = run_case(c, a.iters, a.fresh, hammer)
            if rec is None:
                print({"case": c.name, "skipped": "unsupported"}, flush=True)
                continue
            print(rec, flush=True)
            failed += bool(rec["bad_launches"]) or not rec["inputs_intact"]
        torch.cuda.empty_cache()
    if a.harness:
        rec = harness_qkv(a.harness, hammer)
        print(rec, flush=True)
        failed += bool(rec["bad_count"])
    print("STRESS", "FAILED" if failed else "CLEAN", failed, flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
