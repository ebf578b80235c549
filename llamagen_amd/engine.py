"""DecodeEngine: device state + launch sequence of the GPT decode loop on the HIP library.

Owns everything `Transformer.setup_caches` allocates in the reference (gpt.py:316-330) plus the
MI355X-specific state: fragment-packed weight copies, fragment-packed activation workspaces, KV
slabs [L][B2][H][S8][hd_pad], the device-side (pos, step) counters and the captured hipGraph of
one decode step (replayed N-2 times per generate()).  PyTorch is used for device memory, streams,
the Exp(1) noise draw and graph capture only -- every FLOP is a kernel of liblgen_hip.so.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import _lib as L


def _ceil_div(a, b):
    return (a + b - 1) // b


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> WP[N/16][K/KC][4 g][16 r][EPL] (one-time layout transform)."""
    N, K = w.shape
    epl = 8 if w.dtype == torch.bfloat16 else 4
    kc = 4 * epl
    assert N % 16 == 0 and K % kc == 0, (N, K)
    return w.view(N // 16, 16, K // kc, 4, epl).permute(0, 2, 3, 1, 4).contiguous()


def pack_act(x: torch.Tensor, mts: int) -> torch.Tensor:
    """[M, K] row-major -> XP[K/KC][MTs][4 g][16 r][EPL] (rows padded with zeros)."""
    M, K = x.shape
    epl = 8 if x.dtype == torch.bfloat16 else 4
    kc = 4 * epl
    assert K % kc == 0 and M <= mts * 16
    xp = torch.zeros(mts * 16, K, dtype=x.dtype, device=x.device)
    xp[:M] = x
    return xp.view(mts, 16, K // kc, 4, epl).permute(2, 0, 3, 1, 4).contiguous()


def unpack_act(xp: torch.Tensor, M: int) -> torch.Tensor:
    """Inverse of pack_act: XP[K/KC][MTs][4][16][EPL] -> [M, K]."""
    if xp.dim() == 4:  # [K/KC][MTs][64][EPL] workspace view
        xp = xp.view(xp.shape[0], xp.shape[1], 4, 16, xp.shape[3])
    kch, mts, g, r, epl = xp.shape
    return xp.permute(1, 3, 0, 2, 4).reshape(mts * 16, kch * g * epl)[:M].contiguous()


def precompute_freqs_cis_2d(grid_size: int, n_elem: int, base: float, cls_token_num: int) -> torch.Tensor:
    """2-D RoPE table of gpt.py:404-417 ([cls + grid^2, n_elem/2, 2] fp32, conditioning rows zero)."""
    half = n_elem // 2
    freqs = 1.0 / (base ** (torch.arange(0, half, 2)[: (half // 2)].float() / half))
    t = torch.arange(grid_size)
    freqs = torch.outer(t, freqs)
    fg = torch.concat([freqs[:, None, :].expand(-1, grid_size, -1), freqs[None, :, :].expand(grid_size, -1, -1)], dim=-1)
    cache = torch.stack([torch.cos(fg), torch.sin(fg)], dim=-1).flatten(0, 1)
    return torch.cat([torch.zeros(cls_token_num, n_elem // 2, 2), cache])


class DecodeEngine:
    def __init__(self, model, max_batch: int, S8: int, dtype: torch.dtype):
        cfg = model.config
        if dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError("the HIP engine implements --precision bf16 and none (fp32); fp16 is not built")
        self.lib = L.lib()
        self.dev = model.tok_embeddings.weight.device
        self.dtype = dtype
        self.dt = L.BF16 if dtype == torch.bfloat16 else L.F32
        self.epl = 8 if dtype == torch.bfloat16 else 4
        self.kc = 4 * self.epl
        self.B2, self.S8 = max_batch, S8
        self.L, self.H, self.d = cfg.n_layer, cfg.n_head, cfg.dim
        self.hd = cfg.dim // cfg.n_head
        self.hdp = 64 if self.hd <= 64 else 128
        if self.hd > 128 or self.hd % 4:
            raise NotImplementedError(f"head_dim {self.hd} unsupported")
        self.F = model.layers[0].feed_forward.hidden_dim
        self.V = cfg.vocab_size
        self.eps = cfg.norm_eps
        self.T = cfg.cls_token_num
        for n in (self.d, self.F):
            if n % 32:
                raise NotImplementedError("model dims must be multiples of 32")
        if self.V % 16:
            raise NotImplementedError("vocab_size must be a multiple of 16")
        mts = _ceil_div(max_batch, 16)
        if mts > 8:
            mts = _ceil_div(mts, 8) * 8
        elif mts > 4:
            mts = 8
        elif mts == 3:
            mts = 4
        self.MTs = mts
        self.mt = min(mts, 8) if mts % 8 == 0 else mts  # m-tiles per workgroup (1, 2, 4 or 8)
        dev, dt = self.dev, dtype
        z = lambda *s, dtype=dt: torch.zeros(*s, dtype=dtype, device=dev)
        # KV slabs (gpt.py:170-185); hd padded to 64/128 so a key row is a power-of-two lane group
        self.k_cache = z(self.L, max_batch, self.H, S8, self.hdp)
        self.v_cache = z(self.L, max_batch, self.H, S8, self.hdp)
        self.causal_mask = torch.tril(torch.ones(S8, S8, dtype=torch.bool, device=dev)).unsqueeze(0).repeat(max_batch, 1, 1)
        grid = int(cfg.block_size ** 0.5)
        self.freqs_cis = precompute_freqs_cis_2d(grid, self.hd, cfg.rope_base, self.T).to(dev).contiguous()
        # activation workspaces (fragment-packed unless noted)
        self.hp = z(self.d // self.kc, mts, 64, self.epl)
        self.xnp = z(self.d // self.kc, mts, 64, self.epl)
        self.ap = z(self.d // self.kc, mts, 64, self.epl)
        self.gp = z(self.F // self.kc, mts, 64, self.epl)
        self.qbuf = z(mts * 16, self.H, self.hdp)            # row-major q (zero pad lanes stay zero)
        self.logits = z(mts * 16, self.V)                    # row-major, storage dtype (gpt.py:368)
        self.noise = z(max_batch, self.V, dtype=torch.float32)
        self.cur_tok = z(mts * 16, dtype=torch.int32)
        self.seq = z(max_batch, S8 + 8, dtype=torch.int32)
        self.state = z(2, dtype=torch.int32)                 # [pos, step]
        self.use_mask = False  # True once causal_mask deviates from pure causal (t2i emb_masks)
        self._graphs = {}
        self._prof = None
        self._pack(model)

    # ---- weights --------------------------------------------------------------------------
    def _sig(self, model):
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def _pack(self, model):
        dt = self.dtype
        cast = lambda p: p.detach().to(dt)
        self.layers = []
        for blk in model.layers:
            at, ff = blk.attention, blk.feed_forward
            w1p, w3p = pack_weight(cast(ff.w1.weight)), pack_weight(cast(ff.w3.weight))
            self.layers.append(dict(
                an=cast(blk.attention_norm.weight).contiguous(), fn=cast(blk.ffn_norm.weight).contiguous(),
                wqkv=pack_weight(cast(at.wqkv.weight)), wo=pack_weight(cast(at.wo.weight)),
                w13=torch.stack([w1p, w3p], dim=1).flatten(0, 1).contiguous(), w2=pack_weight(cast(ff.w2.weight))))
        self.norm_w = cast(model.norm.weight).contiguous()
        self.out_w = pack_weight(cast(model.output.weight))
        self.tok_emb = cast(model.tok_embeddings.weight).contiguous()
        if model.model_type == "c2i":
            self.cls_emb = cast(model.cls_embedding.embedding_table.weight).contiguous()
        else:
            self.fc1 = pack_weight(cast(model.cls_embedding.cap_proj.fc1.weight))
            self.fc2 = pack_weight(cast(model.cls_embedding.cap_proj.fc2.weight))
            self.cap_hidden = model.cls_embedding.cap_proj.fc1.weight.shape[0]
        self._wsig = self._sig(model)
        self._graphs = {}

    def compatible(self, model, max_batch, S8, dtype) -> bool:
        return (self.B2 == max_batch and self.S8 == S8 and self.dtype == dtype
                and self.dev == model.tok_embeddings.weight.device and self._wsig == self._sig(model))

    def reset(self, max_batch: int):
        """What re-running setup_caches means in the reference: fresh (zero) caches, position 0."""
        self.k_cache.zero_()
        self.v_cache.zero_()
        self.state.zero_()
        self.use_mask = False
        self.causal_mask.copy_(torch.tril(torch.ones(self.S8, self.S8, dtype=torch.bool, device=self.dev)))

    # ---- tile heuristics --------------------------------------------------------------------
    def _tiles(self, N: int, K: int, swiglu: bool = False):
        """(mt, nt, kw): workgroups ~ fill 256 CUs, >= 2 k-chunks per wave, <= 8 K-splitting waves."""
        env = os.environ.get("LGEN_TILES")
        ntiles = N // 16
        nt = 2 if (swiglu or ntiles >= 512) else 1
        if ntiles % nt:
            nt = 1
        blocks = ntiles // nt
        kch = K // self.kc
        kw = max(1, min(8, round(2048 / blocks), kch // 2))
        mt = self.mt
        if mt == 8 and nt > 2:
            nt = 2
        if mt * nt >= 16:
            kw = min(kw, 8)
        if env:
            _, nt, kw = [int(v) for v in env.split(",")]
        return mt, nt, kw

    # ---- launches -----------------------------------------------------------------------------
    def gemm(self, wp, xp, out, M, mts, N, K, epi, tiles=None):
        mt, nt, kw = tiles or self._tiles(N, K, epi == L.EPI_SWIGLU)
        if mts % mt:
            mt = math.gcd(mts, mt)
        L.check(self.lib.lgen_gemm(L.ptr(wp), L.ptr(xp), L.ptr(out), M, mts, N, K, epi, self.dt, mt, nt, kw, L.stream()),
                "lgen_gemm")

    def _layers_and_logits(self, want_logits: bool = True):
        lib, st, dt, M, mts = self.lib, L.stream(), self.dt, self.B2, self.MTs
        d, F, H, hd, hdp, S8 = self.d, self.F, self.H, self.hd, self.hdp, self.S8
        pos_ptr = self.state.data_ptr()
        tq = self._tiles(3 * d, d)
        pm = self.causal_mask if self.use_mask else None
        for i, w in enumerate(self.layers):
            L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(w["an"]), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
            L.check(lib.lgen_gemm_qkv_rope(L.ptr(w["wqkv"]), L.ptr(self.xnp), L.ptr(self.qbuf), L.ptr(self.k_cache[i]),
                                           L.ptr(self.v_cache[i]), L.ptr(self.freqs_cis), pos_ptr, M, mts, d, H, hd, hdp,
                                           S8, dt, tq[0], tq[1], tq[2], st), "gemm_qkv_rope")
            if self._prof is not None:  # bench.py roofline leg: HIP events on the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            L.check(lib.lgen_attn_decode(L.ptr(self.qbuf), L.ptr(self.k_cache[i]), L.ptr(self.v_cache[i]), L.ptr(self.ap),
                                         pos_ptr, L.ptr(pm), M, mts, H, hd, hdp, S8, dt, st), "attn_decode")
            if self._prof is not None:
                e1.record()
                self._prof["events"].append((e0, e1))
            self.gemm(w["wo"], self.ap, self.hp, M, mts, d, d, L.EPI_RES)
            L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(w["fn"]), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
            self.gemm(w["w13"], self.xnp, self.gp, M, mts, 2 * F, d, L.EPI_SWIGLU)
            self.gemm(w["w2"], self.gp, self.hp, M, mts, d, F, L.EPI_RES)
        if want_logits:
            L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(self.norm_w), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
            self.gemm(self.out_w, self.xnp, self.logits, M, mts, self.V, d, L.EPI_ROWS)

    def _embed(self, table, idx):
        L.check(self.lib.lgen_embed_pack(L.ptr(table), L.ptr(idx), L.ptr(self.hp), self.B2, self.MTs, self.d,
                                         table.shape[0], self.dt, L.stream()), "embed_pack")

    def _sample(self, B, sp, advance=1):
        greedy = 0 if sp["sample_logits"] else 1
        if not greedy:
            self.noise[:B].exponential_(1.0) if sp.get("_noise") is None else self.noise[:B].copy_(sp["_noise"])
        L.check(self.lib.lgen_sample(L.ptr(self.logits), L.ptr(self.noise), L.ptr(self.cur_tok), L.ptr(self.seq),
                                     L.ptr(self.state), B, self.V, self.seq.shape[1], 1 if sp["use_cfg"] else 0,
                                     float(sp["cfg_scale"]), int(sp["cfg_interval"]), float(sp["temperature"]),
                                     int(sp["top_k"]), float(sp["top_p"]), greedy, advance, self.dt, L.stream()),
                "lgen_sample")

    def decode_step(self, B, sp):
        """One KV-cached decode step: embed(cur_tok) -> L blocks -> logits -> sample -> advance."""
        self._embed(self.tok_emb, self.cur_tok)
        self._layers_and_logits()
        self._sample(B, sp)

    # ---- t2i prefix (CaptionEmbedder, gpt.py:110-131) ------------------------------------------
    def caption_embed(self, cond: torch.Tensor) -> torch.Tensor:
        """cond [B2, T, caption_dim] -> [B2, T, d] through the HIP GEMMs (fc1+gelu_tanh, fc2)."""
        B2, T, C = cond.shape
        x = cond.reshape(B2 * T, C).to(self.dtype)
        mts = _ceil_div(B2 * T, 16)
        mts = _ceil_div(mts, 8) * 8 if mts > 4 else (4 if mts == 3 else mts)
        xp = pack_act(x, mts)
        hid = torch.zeros(self.cap_hidden // self.kc, mts, 64, self.epl, dtype=self.dtype, device=self.dev)
        out = torch.zeros(mts * 16, self.d, dtype=self.dtype, device=self.dev)
        mt = 8 if mts % 8 == 0 else mts
        self.gemm(self.fc1, xp, hid, B2 * T, mts, self.cap_hidden, C, L.EPI_GELU, tiles=(mt, 1, 1))
        self.gemm(self.fc2, hid, out, B2 * T, mts, self.d, self.cap_hidden, L.EPI_ROWS, tiles=(mt, 1, 1))
        return out[: B2 * T].view(B2, T, self.d)

    # ---- the generate() loop ---------------------------------------------------------------
    def generate(self, model, cond_combined, B, max_new_tokens, emb_masks, sp):
        """prefill + (N-1) decode steps; returns int32 [B, N].  sp: sampling parameter dict."""
        N = max_new_tokens
        T = 1 if model.model_type == "c2i" else cond_combined.shape[1]
        self._prof = getattr(model, "_prof", None)
        self.state.zero_()
        if emb_masks is not None:  # generate.py:154-163: fold emb_masks into causal_mask, force the diagonal
            cm = self.causal_mask
            cm[:, :, :T] = cm[:, :, :T] & emb_masks.to(self.dev).ne(0).unsqueeze(1)
            cm |= torch.eye(self.S8, dtype=torch.bool, device=self.dev)
            self.use_mask = True
        noise_seq = sp.pop("_noise_seq", None)

        def nz(i):
            sp["_noise"] = None if noise_seq is None else noise_seq[i].to(self.dev)

        # ---- prefill (generate.py:77-86)
        nz(0)
        if model.model_type == "c2i":
            self._embed(self.cls_emb, cond_combined.to(torch.int32).contiguous())
            self._layers_and_logits()
        else:
            emb = self.caption_embed(cond_combined)
            for t in range(T):  # causal prefix, one position at a time (same math as the batched prefill)
                self.state[0] = t
                self.hp.copy_(pack_act(emb[:, t].contiguous(), self.MTs).view_as(self.hp))
                self._layers_and_logits(want_logits=(t == T - 1))
        self._sample(B, sp)  # -> state = [T, 1]
        # ---- decode (generate.py:105-123)
        key = (B, self.use_mask, sp["use_cfg"], sp["cfg_scale"], sp["cfg_interval"], sp["temperature"], sp["top_k"], sp["top_p"],
               sp["sample_logits"])
        use_graph = noise_seq is None and os.environ.get("LGEN_NO_GRAPH") is None and N > 3
        i = 1
        if use_graph:
            if i < N:  # first decode step eagerly (also warms every kernel before capture)
                nz(i)
                self.decode_step(B, sp)
                i += 1
            g = self._graphs.get(key)
            if g is None:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    self.decode_step(B, sp)
                self._graphs[key] = g
            while i < N:
                g.replay()
                i += 1
        else:
            while i < N:
                nz(i)
                self.decode_step(B, sp)
                i += 1
        return self.seq[:B, :N].clone()

    # ---- Transformer.forward inference branches (gpt.py:347-368) ---------------------------------
    def forward(self, model, idx, cond_idx, input_pos):
        pos = [int(p) for p in input_pos.reshape(-1).tolist()]
        outs = []
        if cond_idx is not None:
            if model.model_type == "c2i":
                embs = None
                rows = cond_idx.reshape(-1).to(torch.int32).contiguous()
            else:
                embs = self.caption_embed(cond_idx)
            nb = cond_idx.shape[0]
        else:
            rows = idx.reshape(-1).to(torch.int32).contiguous()
            embs = None
            nb = idx.shape[0]
        if nb != self.B2:
            raise ValueError(f"batch {nb} != max_batch_size {self.B2} given to setup_caches")
        for j, p in enumerate(pos):
            self.state[0] = p
            if embs is not None:
                self.hp.copy_(pack_act(embs[:, j].contiguous(), self.MTs).view_as(self.hp))
            elif cond_idx is not None:
                self._embed(self.cls_emb, rows)
            else:
                self._embed(self.tok_emb, rows)
            self._layers_and_logits()
            outs.append(self.logits[:nb].float())
        return torch.stack(outs, dim=1)
