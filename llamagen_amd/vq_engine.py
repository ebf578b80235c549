"""VQEngine: launch sequence of the VQ-VAE decode path (and codebook argmin) on the HIP library.

Follows VQModel.decode_code -> decode -> Decoder.forward of the reference
(tokenizer/tokenizer_image/vq_model.py:47-55, 173-194): codebook gather + post_quant_conv, conv_in,
mid (Res, Attn, Res), the up-sampling levels, norm_out/swish/conv_out.  Activations are NHWC fp32 in
HBM; every convolution / bmm is `lgen_conv_igemm` (MFMA, hi/lo-split bf16 operands); GroupNorm is a
statistics pass + a fused apply/swish/split pass.  torch is used for buffers and the one-time weight
re-layout only.
"""
from __future__ import annotations

import os

import torch

from . import _lib as L


def _split_planes(w: torch.Tensor):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


class _ConvW:
    """[Cout, Cin, k, k] fp32 -> (hi, lo) bf16 planes [k*k][Npad][Cin] + fp32 bias."""

    def __init__(self, conv):
        w = conv.weight.detach().float()
        cout, cin, k, _ = w.shape
        blk = 128 if cout >= 128 else (64 if cout >= 64 else 16)
        npad = (cout + blk - 1) // blk * blk
        cpad = (cin + 31) // 32 * 32  # the igemm K-step is 32 input channels (RGB conv_in: 3 -> 32, zero weights)
        wt = torch.zeros(k * k, npad, cpad, device=w.device)
        wt[:, :cout, :cin] = w.permute(2, 3, 0, 1).reshape(k * k, cout, cin)
        self.hi, self.lo = _split_planes(wt)
        self.bias = conv.bias.detach().float().contiguous()
        self.cin, self.cout, self.npad, self.k = cpad, cout, npad, k
        # fragment-packed copy for lgen_conv_fused: [Npad/BN][Cin/32][taps][hi|lo][BN/16][4 g][16 r][8] -- one 1 KiB
        # block is one MFMA A fragment (lane = g*16 + r holds output channel r, input channels g*8..g*8+7)
        bn = 128 if cout >= 128 else (64 if cout > 16 else 16)
        self.fnpad = (cout + bn - 1) // bn * bn
        wf = torch.zeros(k * k, self.fnpad, cpad, device=w.device)
        wf[:, :cout, :cin] = wt[:, :cout, :cin]
        fh, fl = _split_planes(wf)
        pk = lambda q: q.view(k * k, self.fnpad // bn, bn // 16, 16, cpad // 32, 4, 8).permute(1, 4, 0, 2, 5, 3, 6)
        self.frag = torch.stack([pk(fh), pk(fl)], dim=3).contiguous()  # [nb][kc][tap][plane][j][g][r][8]
        self.bn = bn


class _GNW:
    def __init__(self, gn):
        self.gamma = gn.weight.detach().float().contiguous()
        self.beta = gn.bias.detach().float().contiguous()
        self.eps = gn.eps


class _Act:
    """An NHWC fp32 activation plus, when a fused conv produced it, the per-tile (sum, M2) partials of its values
    (the next GroupNorm's statistics without another pass over the tensor) and the width of the map they tile."""
    __slots__ = ("t", "part", "ntiles", "qstride", "width")

    def __init__(self, t, part=None, ntiles=0, qstride=0, width=0):
        self.t, self.part, self.ntiles, self.qstride, self.width = t, part, ntiles, qstride, width


class VQEngine:
    def __init__(self, model):
        self.lib = L.lib()
        if os.environ.get("LGEN_VQ_NT") is not None:  # tuning knob, see lgen_debug_set_vq_nt in lgen.h
            self.lib.lgen_debug_set_vq_nt(int(os.environ["LGEN_VQ_NT"]))
        self.dev = model.post_quant_conv.weight.device
        self.fused = os.environ.get("LGEN_VQ_FUSED", "1") != "0"  # lgen_conv_fused where the shape allows it
        if os.environ.get("LGEN_CF_VARIANT") is not None:  # tuning knob, see lgen_debug_set_conv_fused_variant in lgen.h
            self.lib.lgen_debug_set_conv_fused_variant(int(os.environ["LGEN_CF_VARIANT"]))
        cfg = model.config
        self.n_e, self.e_dim, self.l2 = cfg.codebook_size, cfg.codebook_embed_dim, cfg.codebook_l2_norm
        self._sig_v = self._sig(model)
        dec = model.decoder
        self.num_res = dec.num_resolutions
        self.nrb = dec.num_res_blocks
        C = lambda m: _ConvW(m)
        G = lambda m: _GNW(m)
        res = lambda r: dict(n1=G(r.norm1), c1=C(r.conv1), n2=G(r.norm2), c2=C(r.conv2),
                             nin=C(r.nin_shortcut) if hasattr(r, "nin_shortcut") else None)
        att = lambda a: dict(n=G(a.norm), q=C(a.q), k=C(a.k), v=C(a.v), p=C(a.proj_out))
        self.conv_in = C(dec.conv_in)
        self.mid = [res(dec.mid[0]), att(dec.mid[1]), res(dec.mid[2])]
        self.levels = []
        for blk in dec.conv_blocks:
            self.levels.append(dict(res=[res(r) for r in blk.res], attn=[att(a) for a in blk.attn],
                                    up=C(blk.upsample.conv) if hasattr(blk, "upsample") else None))
        self.norm_out, self.conv_out = G(dec.norm_out), C(dec.conv_out)
        # post_quant_conv 1x1 (Cin = e_dim): handled by the gather kernel
        self.pq_w = model.post_quant_conv.weight.detach().float().reshape(cfg.z_channels, self.e_dim).contiguous()
        self.pq_b = model.post_quant_conv.bias.detach().float().contiguous()
        self.zc = cfg.z_channels
        cb = model.quantize.embedding.weight.detach().float().contiguous()
        self.cbn = torch.empty_like(cb)
        self.esq = torch.empty(self.n_e, device=self.dev)
        L.check(self.lib.lgen_vq_codebook_prep(L.ptr(cb), L.ptr(self.cbn), L.ptr(self.esq), self.n_e, self.e_dim,
                                               1 if self.l2 else 0, L.stream()), "codebook_prep")
        self._cb = cb
        self._enc = None
        self._model_ref = model
        # weight planes / codebook prep were enqueued on the constructing stream; lanes use this engine from their own
        # streams without an event dependency, so finish the one-time setup here
        torch.cuda.current_stream(self.dev).synchronize()

    @staticmethod
    def _sig(model):
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def compatible(self, model) -> bool:
        return self._sig_v == self._sig(model) and self.dev == model.post_quant_conv.weight.device

    # ---- primitive launches ----------------------------------------------------------------
    def _stats(self, x, B, hw, C, eps):
        nchunk = max(1, min((hw + 63) // 64, max(1, 2048 // max(B, 1))))
        ws = torch.empty(B * nchunk * 64, dtype=torch.float64, device=self.dev)
        st = torch.empty(B, 32, 2, dtype=torch.float32, device=self.dev)
        L.check(self.lib.lgen_gn_stats(L.ptr(x), L.ptr(ws), L.ptr(st), B, hw, C, eps, nchunk, L.stream()), "gn_stats")
        return st

    def _split(self, x, B, hw, C, gn=None, swish=False, slack_rows=0):
        n = B * hw * C + slack_rows * C
        hi = torch.empty(n, dtype=torch.bfloat16, device=self.dev)
        lo = torch.empty(n, dtype=torch.bfloat16, device=self.dev)
        mode = (1 if gn is not None else 0) | (2 if swish else 0)
        st = self._stats(x, B, hw, C, gn.eps) if gn is not None else None
        L.check(self.lib.lgen_gn_swish_split(L.ptr(x), L.ptr(st), L.ptr(gn.gamma) if gn else 0, L.ptr(gn.beta) if gn else 0,
                                             L.ptr(hi), L.ptr(lo), B, hw, C, mode, L.stream()), "gn_swish_split")
        return hi, lo

    def _conv(self, planes, cw: _ConvW, B, H, W, upsample=False, res=None, out_nchw=False):
        """H, W = OUTPUT size; upsample: False / True (nearest-2x input) / 2 (stride-2 Downsample conv)."""
        hi, lo = planes
        out = torch.empty(B * H * W * cw.cout, dtype=torch.float32, device=self.dev)
        L.check(self.lib.lgen_conv_igemm(L.ptr(hi), L.ptr(lo), L.ptr(cw.hi), L.ptr(cw.lo), L.ptr(cw.bias), L.ptr(res),
                                         L.ptr(out), B, H, W, cw.cin, cw.cout, cw.npad, cw.k, int(upsample),
                                         1 if out_nchw else 0, 0, 1.0, L.stream()), "conv_igemm")
        return out

    # ---- fused path (lgen_conv_fused): GroupNorm-apply / swish / split on the tile load, statistics in the epilogue --------
    @staticmethod
    def _fusable(H, W):
        return H % 8 == 0 and W >= 8   # tiles are 8 x 16 pixels; a last tile column may be partly outside the map (24-wide maps)

    def _coef(self, x: "_Act", gn: _GNW, B, hw, C):
        """Per-channel (scale, shift) of GroupNorm `gn` on activation x: from the producer's tile partials when it has them,
        else from a statistics pass."""
        coef = torch.empty(B * C * 2, dtype=torch.float32, device=self.dev)
        if x.part is not None:
            L.check(self.lib.lgen_gn_finalize(L.ptr(x.part), 0, L.ptr(gn.gamma), L.ptr(gn.beta), L.ptr(coef), B, C, x.ntiles,
                                              x.qstride, hw, x.width, gn.eps, L.stream()), "gn_finalize")
        else:
            st = self._stats(x.t, B, hw, C, gn.eps)
            L.check(self.lib.lgen_gn_finalize(0, L.ptr(st), L.ptr(gn.gamma), L.ptr(gn.beta), L.ptr(coef), B, C, 0, 0, hw, 0, gn.eps,
                                              L.stream()), "gn_finalize")
        return coef

    def _convf(self, x: "_Act", cw: _ConvW, B, H, W, coef=None, swish=False, upsample=False, res=None, out_nchw=False,
               want_part=True) -> "_Act":
        """H, W = OUTPUT size.  x.t is fp32 NHWC [B][H>>ups][W>>ups][cin]."""
        out = torch.empty(B * H * W * cw.cout, dtype=torch.float32, device=self.dev)
        ntiles = (H // 8) * ((W + 15) // 16)
        part = torch.empty(B * ntiles * (cw.fnpad // 4) * 2, dtype=torch.float32, device=self.dev) if want_part else None
        L.check(self.lib.lgen_conv_fused(L.ptr(x.t), L.ptr(coef), 1 if swish else 0, L.ptr(cw.frag), L.ptr(cw.bias),
                                         L.ptr(res.t if isinstance(res, _Act) else res), L.ptr(out), L.ptr(part), B, H, W, cw.cin,
                                         cw.cout, cw.fnpad, cw.k, 1 if upsample else 0, 1 if out_nchw else 0, L.stream()),
                "conv_fused")
        return _Act(out, part, ntiles, cw.fnpad // 4, W)

    def _res_fused(self, x: "_Act", p, B, H, W) -> "_Act":
        hw, cin = H * W, p["c1"].cin
        h = self._convf(x, p["c1"], B, H, W, coef=self._coef(x, p["n1"], B, hw, cin), swish=True)
        skip = x if p["nin"] is None else self._convf(x, p["nin"], B, H, W, want_part=False)
        return self._convf(h, p["c2"], B, H, W, coef=self._coef(h, p["n2"], B, hw, p["c1"].cout), swish=True, res=skip)

    def _gemm_nt(self, a_planes, b_planes, B, M, N, K, npad, alpha):
        """out[b] = alpha * A[b] (M x K) . Bm[b] (N x K)^T, batched, via the igemm kernel (1x1, W = 1)."""
        out = torch.empty(B * M * N, dtype=torch.float32, device=self.dev)
        L.check(self.lib.lgen_conv_igemm(L.ptr(a_planes[0]), L.ptr(a_planes[1]), L.ptr(b_planes[0]), L.ptr(b_planes[1]), 0, 0,
                                         L.ptr(out), B, M, 1, K, N, npad, 1, 0, 0, N * K, float(alpha), L.stream()), "gemm_nt")
        return out

    # ---- blocks (vq_model.py:299-314, 327-351) ----------------------------------------------
    def _res(self, x, p, B, H, W):
        if isinstance(x, _Act):
            if self.fused and self._fusable(H, W):
                return self._res_fused(x, p, B, H, W)
            return _Act(self._res(x.t, p, B, H, W))
        hw, cin = H * W, p["c1"].cin
        h = self._conv(self._split(x, B, hw, cin, p["n1"], True), p["c1"], B, H, W)
        cmid = p["c1"].cout
        skip = x if p["nin"] is None else self._conv(self._split(x, B, hw, cin), p["nin"], B, H, W)
        return self._conv(self._split(h, B, hw, cmid, p["n2"], True), p["c2"], B, H, W, res=skip)

    def _attn(self, x, p, B, H, W):
        fused = isinstance(x, _Act) and self.fused and self._fusable(H, W)
        if isinstance(x, _Act) and not fused:
            return _Act(self._attn(x.t, p, B, H, W))
        hw, c = H * W, p["q"].cin
        if fused:  # GroupNorm-apply on the tile load of the three 1x1 convs (no activation, vq_model.py:330-334)
            coef = self._coef(x, p["n"], B, hw, c)
            xa, x = x, x.t
            q = self._convf(xa, p["q"], B, H, W, coef=coef, want_part=False).t
            k = self._convf(xa, p["k"], B, H, W, coef=coef, want_part=False).t
            v = self._convf(xa, p["v"], B, H, W, coef=coef, want_part=False).t
        else:
            hn = self._split(x, B, hw, c, p["n"], False)
            q = self._conv(hn, p["q"], B, H, W)
            k = self._conv(hn, p["k"], B, H, W)
            v = self._conv(hn, p["v"], B, H, W)
        blk = 128 if hw % 128 == 0 else (64 if hw % 64 == 0 else 16)
        npad = (hw + blk - 1) // blk * blk
        ld = (hw + 31) // 32 * 32
        qs = self._split(q, B, hw, c)
        ks = self._split(k, B, hw, c, slack_rows=npad)           # rows past hw of the last image are never stored
        s = self._gemm_nt(qs, ks, B, hw, hw, c, npad, float(int(c) ** (-0.5)))
        ph = torch.empty(B * hw * ld, dtype=torch.bfloat16, device=self.dev)
        pl = torch.empty_like(ph)
        L.check(self.lib.lgen_softmax_split(L.ptr(s), L.ptr(ph), L.ptr(pl), B * hw, hw, ld, L.stream()), "softmax_split")
        vh = torch.empty(B * c * ld, dtype=torch.bfloat16, device=self.dev)
        vl = torch.empty_like(vh)
        L.check(self.lib.lgen_split_t(L.ptr(v), L.ptr(vh), L.ptr(vl), B, hw, c, ld, L.stream()), "split_t")
        cblk = 128 if c % 128 == 0 else (64 if c % 64 == 0 else 16)
        o = self._gemm_nt((ph, pl), (vh, vl), B, hw, c, ld, (c + cblk - 1) // cblk * cblk, 1.0)
        if fused:  # proj_out + skip, with the statistics of the result for the next block's norm1
            return self._convf(_Act(o), p["p"], B, H, W, res=xa)
        return self._conv(self._split(o, B, hw, c), p["p"], B, H, W, res=x)

    # ---- public ----------------------------------------------------------------------------------
    def decode_code(self, code_b, shape, channel_first=True):
        if shape is None:
            raise ValueError("decode_code needs shape=[B, C, h, w] (vq_model.py:52-55)")
        if channel_first:
            B, C, h, w = [int(v) for v in shape]
        else:
            B, h, w, C = [int(v) for v in shape]
        idx = code_b.reshape(-1).to(device=self.dev, dtype=torch.int64).contiguous()
        if idx.numel() != B * h * w or C != self.e_dim:
            raise ValueError(f"code_b has {idx.numel()} ids, shape {shape} expects {B * h * w} x {self.e_dim}")
        z = torch.empty(B * h * w * self.zc, dtype=torch.float32, device=self.dev)
        L.check(self.lib.lgen_vq_lookup_pqconv(L.ptr(self.cbn), L.ptr(idx), L.ptr(self.pq_w), L.ptr(self.pq_b), L.ptr(z),
                                               B * h * w, self.n_e, self.e_dim, self.zc, L.stream()), "lookup_pqconv")
        return self._decoder(z, B, h, w)

    def decode(self, quant):
        """quant: fp32 [B, e_dim, h, w] (already-quantised latents) -> post_quant_conv -> Decoder."""
        B, C, h, w = quant.shape
        zz = quant.to(self.dev, torch.float32).permute(0, 2, 3, 1).reshape(-1, C).contiguous()  # per-pixel "codebook"
        idx = torch.arange(B * h * w, dtype=torch.int64, device=self.dev)
        z = torch.empty(B * h * w * self.zc, dtype=torch.float32, device=self.dev)
        L.check(self.lib.lgen_vq_lookup_pqconv(L.ptr(zz), L.ptr(idx), L.ptr(self.pq_w), L.ptr(self.pq_b), L.ptr(z), B * h * w,
                                               B * h * w, C, self.zc, L.stream()), "pqconv")
        return self._decoder(z, B, h, w)

    def _decoder(self, z, B, H, W):
        """Decoder.forward, vq_model.py:173-194; z NHWC [B*H*W*z_channels]."""
        if self.fused and self._fusable(H, W):
            x = self._convf(_Act(z), self.conv_in, B, H, W)
        else:
            x = _Act(self._conv(self._split(z, B, H * W, self.zc), self.conv_in, B, H, W))
        x = self._res(x, self.mid[0], B, H, W)
        x = self._attn(x, self.mid[1], B, H, W)
        x = self._res(x, self.mid[2], B, H, W)
        for li, lv in enumerate(self.levels):
            for bi, rp in enumerate(lv["res"]):
                x = self._res(x, rp, B, H, W)
                if lv["attn"]:
                    x = self._attn(x, lv["attn"][bi], B, H, W)
            if lv["up"] is not None:
                c = lv["up"].cin
                if self.fused and self._fusable(2 * H, 2 * W):
                    x = self._convf(x, lv["up"], B, 2 * H, 2 * W, upsample=True)
                else:
                    x = _Act(self._conv(self._split(x.t, B, H * W, c), lv["up"], B, 2 * H, 2 * W, upsample=True))
                H, W = 2 * H, 2 * W
        c = self.conv_out.cin
        if self.fused and self._fusable(H, W):
            out = self._convf(x, self.conv_out, B, H, W, coef=self._coef(x, self.norm_out, B, H * W, c), swish=True,
                              out_nchw=True, want_part=False).t
        else:
            out = self._conv(self._split(x.t, B, H * W, c, self.norm_out, True), self.conv_out, B, H, W, out_nchw=True)
        return out.view(B, self.conv_out.cout, H, W)

    def argmin(self, z):
        """Nearest codebook entry per latent vector (vq_model.py:215-232): z [B, e_dim, h, w] -> int64 [B*h*w]."""
        B, C, h, w = z.shape
        zc = z.to(self.dev, torch.float32).contiguous()
        out = torch.empty(B * h * w, dtype=torch.int64, device=self.dev)
        L.check(self.lib.lgen_vq_argmin(L.ptr(zc), L.ptr(self.cbn), L.ptr(self.esq), L.ptr(out), B * h * w, h * w, self.n_e, C,
                                        1 if self.l2 else 0, L.stream()), "vq_argmin")
        return out

    # ---- encoder (vq_model.py:64-124, 41-45) ----------------------------------------------------------
    def _encoder_weights(self):
        if self._enc is None:
            model = self._model_ref
            enc = model.encoder
            C, G = _ConvW, _GNW
            res = lambda r: dict(n1=G(r.norm1), c1=C(r.conv1), n2=G(r.norm2), c2=C(r.conv2),
                                 nin=C(r.nin_shortcut) if hasattr(r, "nin_shortcut") else None)
            att = lambda a: dict(n=G(a.norm), q=C(a.q), k=C(a.k), v=C(a.v), p=C(a.proj_out))
            self._enc = dict(
                conv_in=C(enc.conv_in),
                levels=[dict(res=[res(r) for r in blk.res], attn=[att(a) for a in blk.attn],
                             down=C(blk.downsample.conv) if hasattr(blk, "downsample") else None) for blk in enc.conv_blocks],
                mid=[res(enc.mid[0]), att(enc.mid[1]), res(enc.mid[2])],
                norm_out=G(enc.norm_out), conv_out=C(enc.conv_out), quant=C(model.quant_conv))
        return self._enc

    def encode(self, x):
        """VQModel.encode (eval): x fp32 [B, 3, H, W] -> (quant [B, e_dim, h, w], (None, None, None, 0),
        (None, None, indices int64 [B*h*w])) like vq_model.py:41-45 / 254-259 with self.training False."""
        E = self._encoder_weights()
        B, Cx, H, W = x.shape
        down = len(E["levels"]) - 1
        if H % (1 << down) or W % (1 << down):
            raise ValueError(f"image size {H}x{W} must be a multiple of {1 << down}")
        cin = E["conv_in"].cin
        xp = torch.zeros(B, H, W, cin, dtype=torch.float32, device=self.dev)  # NHWC, RGB padded to one 32-channel K-step
        xp[..., :Cx] = x.to(self.dev, torch.float32).permute(0, 2, 3, 1)
        fz = lambda hh, ww: self.fused and self._fusable(hh, ww)
        if fz(H, W):
            h = self._convf(_Act(xp.reshape(-1)), E["conv_in"], B, H, W)
        else:
            h = _Act(self._conv(self._split(xp.reshape(-1), B, H * W, cin), E["conv_in"], B, H, W))
        for lv in E["levels"]:
            for bi, rp in enumerate(lv["res"]):
                h = self._res(h, rp, B, H, W)
                if lv["attn"]:
                    h = self._attn(h, lv["attn"][bi], B, H, W)
            if lv["down"] is not None:  # stride-2 Downsample conv (vq_model.py:389-393): implicit-GEMM path
                c = lv["down"].cin
                h = _Act(self._conv(self._split(h.t, B, H * W, c), lv["down"], B, H // 2, W // 2, upsample=2))
                H, W = H // 2, W // 2
        h = self._res(h, E["mid"][0], B, H, W)
        h = self._attn(h, E["mid"][1], B, H, W)
        h = self._res(h, E["mid"][2], B, H, W)
        c = E["conv_out"].cin
        zc = E["quant"].cin
        if fz(H, W):
            h = self._convf(h, E["conv_out"], B, H, W, coef=self._coef(h, E["norm_out"], B, H * W, c), swish=True, want_part=False)
            z = self._convf(h, E["quant"], B, H, W, out_nchw=True, want_part=False).t.view(B, self.e_dim, H, W)
        else:
            h = self._conv(self._split(h.t, B, H * W, c, E["norm_out"], True), E["conv_out"], B, H, W)
            z = self._conv(self._split(h, B, H * W, zc), E["quant"], B, H, W, out_nchw=True).view(B, self.e_dim, H, W)
        idx = self.argmin(z)
        emb = self.cbn if self.l2 else self._cb
        quant = emb[idx].view(B, H, W, self.e_dim).permute(0, 3, 1, 2).contiguous()  # gather of codebook rows (no arithmetic)
        self.last_latent = z
        return quant, (None, None, None, 0), (None, None, idx)
