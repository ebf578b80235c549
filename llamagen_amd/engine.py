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
import weakref
from typing import Optional

import torch

from . import _lib as L


def _ceil_div(a, b):
    return (a + b - 1) // b


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> WP[N/16][K/KC][4 g][16 r][EPL] (one-time layout transform)."""
    N, K = w.shape
    epl = 4 if w.dtype == torch.float32 else 8
    kc = 4 * epl
    assert N % 16 == 0 and K % kc == 0, (N, K)
    return w.view(N // 16, 16, K // kc, 4, epl).permute(0, 2, 3, 1, 4).contiguous()


def pack_act(x: torch.Tensor, mts: int) -> torch.Tensor:
    """[M, K] row-major -> XP[K/KC][MTs][4 g][16 r][EPL] (rows padded with zeros)."""
    M, K = x.shape
    epl = 4 if x.dtype == torch.float32 else 8
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


# Big-M tile family (csrc/gemm_tile.hip): workgroup shapes (wm, wn, mtv, ntv, kb, stages, lw) per decode GEMM, keyed by the smallest
# MTs (rows / 16) they serve.  Per-CU tile area ~ rows x N / 256 CUs, as square as the wave arrangement allows (the feed per CU is the
# tile perimeter x K); measured on MI355X with tools/gemm_tile_sweep.py (gpurun_out/tile_sweep2/3.log, GPT-L, us per launch,
# skinny -> tile): 256 rows wqkv 10.2 -> 9.8, wo 4.85 -> 4.1, w1||w3 13.1 -> 10.6, w2 7.8 -> 7.7, lm_head 23.2 -> 20.9; 512 rows
# 16.3 -> 12.9, 7.7 -> 5.1, 20.3 -> 18.9, 9.5 -> 7.9, 39.0 -> 37.8; 1024 rows 51.6 -> 22.1, 13.0 -> 6.6, 34.1 -> 35.0, 16.9 -> 11.5,
# 71.1 -> 68.8.  The 256- and 640-row tables are what tests/test_gpu_headline.py holds to the oracle end to end and what bench.py
# accepts as a tested schedule; every shape of every table is held to the oracle at kernel level (and all shapes agree bit for bit).
TILE_SCHEDULES = {
    16: {"qkv": (4, 1, 1, 3, 4, 4, 4), "wo": (2, 2, 1, 1, 4, 4, 4), "w13": (4, 1, 1, 6, 4, 3, 4), "w2": (2, 2, 1, 1, 4, 4, 4),
         "head": (4, 1, 2, 8, 2, 4, 4)},
    32: {"qkv": (4, 1, 1, 6, 4, 3, 4), "wo": (2, 2, 2, 1, 4, 4, 4), "w13": (4, 1, 2, 8, 2, 4, 4), "w2": (2, 2, 2, 1, 4, 4, 4),
         "head": (4, 1, 2, 8, 2, 4, 4)},
    # 640 rows (`bench.py --steps 20`: two chains of ten batches), gpurun_out/tile_sweep4.log: a launch is whole ROUNDS of <= 256
    # workgroups, so the shapes are the ones that fill one round -- wqkv 240 workgroups 16.0 us (22.1 with the 512-row shape: 320
    # workgroups = two rounds), wo 160 / 6.3 (8.1), w1||w3 220 / 18.2, w2 160 / 10.9 (14.6), lm_head 640 / 50.3
    # (w1||w3 and lm_head: eight consumer waves x one m-tile -- two per SIMD, one's RMSNorm VALU work under the other's MFMAs:
    # 18.2 against 19.2 us and 50.3 against 52.0 for four waves x two m-tiles, gpurun_out/tile_sweep5.log)
    40: {"qkv": (4, 1, 1, 8, 2, 4, 4), "wo": (2, 2, 2, 2, 4, 4, 4), "w13": (8, 1, 1, 8, 2, 4, 4), "w2": (2, 2, 2, 2, 4, 4, 4),
         "head": (8, 1, 1, 8, 2, 4, 4)},
    64: {"qkv": (4, 1, 2, 8, 2, 4, 4), "wo": (2, 2, 2, 2, 4, 4, 4), "w13": (4, 1, 2, 8, 2, 4, 4), "w2": (2, 2, 2, 2, 4, 4, 4),
         "head": (4, 1, 2, 8, 2, 4, 4)},
}
TILE_SCHEDULE_EXACT = (40,)        # keys that serve exactly that MTs; the others serve every MTs from the key up to the next one
TESTED_TILE_SCHEDULES = (16, 40)   # keys of TILE_SCHEDULES with an end-to-end oracle test (tests/test_gpu_headline.py)
TABLE_MODEL = (1024, 2816, 16384)   # (dim, ffn hidden, vocab) of GPT-L: the model TILE_SCHEDULES was measured on

# The other registry models that bench.py runs with chains of >= 256 rows: the shapes the round-4 on-device search settled on
# (profiles/r04_bench_config{3,4}.json, `schedule`), now PINNED.  Round 4 timed every shape (and the skinny kernel) at first use in
# the user's process and kept whichever was faster at that moment: the tile shapes agree bit for bit, but tile and skinny differ in
# accumulation order and in the SwiGLU's exp / reciprocal, so the family -- and with it bf16 logits -- depended on timing noise
# (ADVICE round 4).  The family now depends on shapes only: tile wherever a table names a shape the library takes.
MODEL_TILE_SCHEDULES = {
    TABLE_MODEL: TILE_SCHEDULES,
    # Round 6: chains of >= 512 rows for the wide models (bench.py --config 3 / 4: two chains of eight batches of 32 / four batches of
    # 64) with shapes MEASURED per model and width (tools/gemm_tile_sweep.py, profiles/r06_tile_sweep_{xxl,3b,xl}.log, us per launch):
    # GPT-XXL 512 rows wqkv 24.0 (28.9 with the 256-row shape), w1||w3 26.1, wo 8.4, w2 19.6, lm_head 50.5; GPT-3B 512 rows wqkv 99.4,
    # w1||w3 140.7 (153.3), wo 23.7 (32.8), w2 53.6 (75.0), lm_head 109.3 (114.8); GPT-XL 256 rows (config 5: two chains of eight
    # batches of 16) wqkv 12.8 (skinny 15.4), wo 5.6 (7.3), w1||w3 14.8 (20.0), w2 13.3 (12.6), lm_head 24.5 (27.1).
    (1536, 4096, 16384): {   # GPT-XXL (config 3)
        16: {"qkv": (4, 1, 1, 8, 2, 4, 4), "wo": (2, 2, 2, 2, 4, 4, 4), "w13": (4, 1, 2, 8, 2, 4, 4), "w2": (2, 2, 2, 2, 4, 4, 4),
             "head": (4, 1, 2, 8, 2, 4, 4)},
        # wqkv: four consumer waves x two m-tiles instead of eight x one -- 25.5 against 24.0 us alone, 47.5 against 45.6 / 46.0 img/s in
        # the two-chain bench (same box, profiles/r06_c3_c5_shape_ab.log; the other candidates of that log lose or tie)
        32: {"qkv": (4, 1, 2, 6, 2, 4, 4), "wo": (2, 2, 2, 2, 4, 4, 4), "w13": (4, 1, 2, 8, 2, 4, 4), "w2": (2, 2, 2, 2, 4, 4, 4),
             "head": (4, 1, 2, 8, 2, 4, 4)}},
    (3200, 8704, 16384): {   # GPT-3B (config 4)
        16: {"qkv": (8, 1, 1, 6, 2, 4, 4), "wo": (2, 2, 2, 2, 4, 4, 4), "w13": (4, 1, 1, 6, 2, 4, 4), "w2": (2, 2, 2, 2, 4, 4, 4),
             "head": (8, 1, 1, 8, 2, 4, 4)},
        # 512 rows, statistics grouped (lgen_ssq_group4): wqkv 128 x 160 tiles = 240 workgroups, ONE round of the 256 CUs, 53.3 us
        # (75.3 with 128 x 96 = 400 workgroups = two rounds; 98.9 before the grouping: profiles/r06_tile_sweep2_3b.log).  w1||w3 keeps
        # 128 x 96: the 128 x 160 shape is faster alone (90.1 against 102.9 us) and SLOWER in the two-chain bench -- same box,
        # alternating (profiles/r06_c4_shape_ab.log): both new 31.4 / 31.4, both old 36.0 / 35.8, new wqkv only 37.2, new w1||w3 only 31.3 img/s
        32: {"qkv": (4, 1, 2, 10, 2, 4, 4), "wo": (2, 2, 4, 2, 2, 4, 4), "w13": (4, 1, 2, 6, 2, 4, 4), "w2": (2, 2, 4, 2, 2, 4, 4),
             "head": (4, 1, 2, 8, 2, 4, 4)}},
    (1280, 3584, 16384): {   # GPT-XL (config 5, t2i)
        16: {"qkv": (4, 1, 1, 4, 4, 4, 4), "wo": (2, 2, 2, 1, 4, 4, 4), "w13": (4, 1, 1, 8, 2, 4, 4), "w2": (2, 2, 2, 1, 4, 4, 4),
             "head": (8, 1, 1, 8, 2, 4, 4)},
        # 384 rows (two chains of twelve batches of 16, what bench.py --config 5 runs): 64 x 64 wqkv tiles would be 360 workgroups (two
        # rounds of the 256 CUs), 64 x 128 w1||w3 tiles 336; 64 x 96 (240) and 128 x 128 (168) fill one -- 21.6 / 21.45 against 21.16 /
        # 21.25 img/s in the two-chain bench, same box, alternating (profiles/r06_c5_shape_ab2.log)
        24: {"qkv": (4, 1, 1, 6, 4, 3, 4), "wo": (2, 2, 2, 1, 4, 4, 4), "w13": (4, 1, 2, 8, 2, 4, 4), "w2": (2, 2, 2, 1, 4, 4, 4),
             "head": (8, 1, 1, 8, 2, 4, 4)}},
}
# (dim, F, V) -> keys of its table that an end-to-end oracle test runs (tests/test_gpu_headline.py:
# test_config{3,4}_..._shapes_bf16_vs_oracle[128] / [192]: 256- / 384-row chains; test_config{3,4,5}_wide_chain_shapes_bf16_vs_oracle:
# the 512- / 512- / 256-row chains bench.py runs since round 6)
TESTED_MODEL_SCHEDULES = {TABLE_MODEL: TESTED_TILE_SCHEDULES, (1536, 4096, 16384): (16, 32), (3200, 8704, 16384): (16, 32),
                          (1280, 3584, 16384): (16, 24)}


def tile_schedule_key(mts: int, table=None):
    """Key of a tile table (default TILE_SCHEDULES) that serves a chain of `mts` m-tiles (None: below 256 rows, the skinny kernels)."""
    table = TILE_SCHEDULES if table is None else table
    exact = TILE_SCHEDULE_EXACT if table is TILE_SCHEDULES else ()
    if mts in exact:
        return mts
    keys = [k for k in table if k not in exact and k <= mts]
    return max(keys) if keys else None


# Every workgroup shape the library instantiates (csrc/gemm_tile.hip GT_SHAPES_NORM / GT_SHAPES_PLAIN): the candidates
# of the opt-in on-device shape search below (LGEN_TILE_AUTOTUNE=1).  All shapes of the family produce bit-identical results (one
# wave accumulates an output element over k in order, whatever the tiling: tests/test_gpu_headline.py), and the search never leaves
# the family, so it only ever changes the speed.
TILE_SHAPES_NORM = ((4, 1, 1, 3, 4, 4, 4), (4, 1, 1, 4, 4, 4, 4), (4, 1, 1, 6, 2, 4, 4), (4, 1, 1, 6, 4, 3, 4), (4, 1, 1, 8, 2, 4, 4),
                    (4, 1, 2, 4, 2, 4, 4), (4, 1, 2, 6, 2, 4, 4), (4, 1, 2, 8, 2, 4, 4), (4, 1, 1, 2, 4, 4, 4), (8, 1, 1, 8, 2, 4, 4),
                    (8, 1, 1, 6, 2, 4, 4), (8, 1, 1, 4, 2, 4, 4), (4, 1, 2, 10, 2, 4, 4))
TILE_SHAPES_PLAIN = ((2, 2, 1, 1, 4, 4, 4), (2, 2, 1, 2, 4, 4, 4), (2, 2, 2, 1, 4, 4, 4), (2, 2, 2, 2, 4, 4, 4), (2, 2, 2, 2, 2, 4, 4),
                     (4, 1, 1, 2, 4, 4, 4), (2, 2, 4, 1, 4, 4, 4), (2, 2, 4, 2, 2, 4, 4))
_TUNED = {}   # (device index, dim, F, V, n_head, MTs, per-row positions?) -> {kind: shape | None}: one search per process, shared by every lane


class PackedWeights:
    """MFMA-fragment-packed copies of a Transformer's parameters (see lgen.h for the layouts)."""

    def __init__(self, model, dtype, sig):
        cast = lambda p: p.detach().to(dtype)
        self.sig, self.dtype = sig, dtype
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
        # The packed copies are built by asynchronous kernels on the constructing stream and then shared by every lane
        # (other streams, no event dependency): finish them here, once, so that no lane can read a half-built copy.
        if self.tok_emb.is_cuda:
            torch.cuda.current_stream(self.tok_emb.device).synchronize()


_PACKED = weakref.WeakKeyDictionary()  # tok_embeddings module (shared by lane views) -> PackedWeights


class DecodeEngine:
    def __init__(self, model, max_batch: int, S8: int, dtype: torch.dtype):
        cfg = model.config
        if dtype not in (torch.bfloat16, torch.float32, torch.float16):
            raise NotImplementedError(f"the HIP engine implements --precision bf16, fp16 and none (fp32), not {dtype}")
        self.lib = L.lib()
        # decode-attention kernel variant: -1 = the library's choice by shape (lgen.h); LGEN_ATTN_VARIANT=n pins one (development)
        self.attn_variant = int(os.environ.get("LGEN_ATTN_VARIANT", "-1"))
        self.dev = model.tok_embeddings.weight.device
        self.dtype = dtype
        self.dt = {torch.bfloat16: L.BF16, torch.float32: L.F32, torch.float16: L.F16}[dtype]
        self.epl = 4 if dtype == torch.float32 else 8
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
        self.model_type = model.model_type
        for n in (self.d, self.F):
            if n % 32:
                raise NotImplementedError("model dims must be multiples of 32")
        if dtype != torch.float32 and self.d > 4096 and not (self.d // 32 % 8 == 0 and 3 <= self.d // 32 // 8 <= 6):
            # the stand-alone 16-bit RMSNorm kernel holds a row in registers up to 4096 columns (lgen_rmsnorm); every registry
            # model is <= 3200 wide -- say so here instead of failing in the middle of a decode chain
            raise NotImplementedError(f"dim {self.d} > 4096 with {dtype}: the stand-alone RMSNorm kernel covers 16-bit rows up to 4096 "
                                      "columns (use --precision none, or a model width the fused-norm GEMMs take)")
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
        self.mt = min(mts, 4)  # m-tiles per workgroup (1, 2 or 4; MTs is 1, 2, 4 or a multiple of 8)
        dev, dt = self.dev, dtype
        z = lambda *s, dtype=dt: torch.zeros(*s, dtype=dtype, device=dev)
        # KV slabs (gpt.py:170-185); hd padded to 64/128 so a key row is a power-of-two lane group.  Two separate
        # slabs by default; LGEN_KV_INTERLEAVE=1 puts the K and V row of a slot next to each other
        # ([.., S8, 2, hdp], row stride 2*hdp: one HBM stream per (b, h) instead of two) -- measured 1.4 % SLOWER
        # in the decode attention at kv_len 576 on MI355X, so it stays an option of the kernels' row-stride argument.
        if os.environ.get("LGEN_KV_INTERLEAVE") == "1":
            self.kv_slab = z(self.L, max_batch, self.H, S8, 2, self.hdp)
            self.k_cache, self.v_cache = self.kv_slab[..., 0, :], self.kv_slab[..., 1, :]
            self.kvs = 2 * self.hdp
        else:
            # Rows are packed at hd rounded up to 8 elements (one 16-byte piece), not at the lane group hdp (round 5): GPT-3B's
            # head_dim 100 -> 104 elements = 13 x 16 B per key instead of 128, i.e. 19 % fewer KV bytes for the HBM-bound decode
            # attention (the reference's own serving fork pads to 112, autoregressive/serve/gpt_model.py:208-220).  The kernels
            # still read hdp elements per key: the lanes past the row see the first bytes of the NEXT row, which meet q's zero
            # pad lanes (QK^T) or output elements >= hd that are never stored (PV); the slack behind the last row keeps those
            # reads inside the allocation (zeros: finite).  LGEN_KV_PACK=0 keeps rows of hdp elements.
            self.kvs = _ceil_div(self.hd, self.epl) * self.epl if os.environ.get("LGEN_KV_PACK", "1") != "0" else self.hdp
            n = self.L * max_batch * self.H * S8 * self.kvs
            self._k_flat, self._v_flat = z(n + self.hdp), z(n + self.hdp)
            self.k_cache = self._k_flat[:n].view(self.L, max_batch, self.H, S8, self.kvs)
            self.v_cache = self._v_flat[:n].view(self.L, max_batch, self.H, S8, self.kvs)
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
        if self.d // 16 > L.SSQ_STRIDE:
            raise ValueError(f"dim {self.d} exceeds the fused-RMSNorm statistics row ({L.SSQ_STRIDE} x 16 columns)")
        self.ssq = z(mts * 16, L.SSQ_STRIDE, dtype=torch.float32)  # per row: d/16 partial sums of squares (fused RMSNorm)
        self.ssq_parts = 0
        # round 6: rows of `parts` partials that the consumers' fast path does not take (parts % 16 != 0 or > 128 -- GPT-3B: 200) are
        # grouped ONCE per consumer launch into their 4 lane-group sums (lgen_ssq_group4 -> ssq_g, parts 4: the same bits) instead
        # of once per consumer workgroup (profiles/r06_tile_ablations_3b.log: 25-38 us of a 99-140 us launch)
        self.ssq_g = z(mts * 16, L.SSQ_STRIDE, dtype=torch.float32)
        self.noise = None            # [N][B][V] fp32 Exp(1) draws of one generate() (allocated on demand)
        self.cur_tok = z(mts * 16, dtype=torch.int32)
        self.seq = z(max_batch, S8 + 8, dtype=torch.int32)
        self.state = z(2, dtype=torch.int32)                 # [pos, step]
        # The decode / prefill attention consults `causal_mask` (the tensor Transformer.setup_caches exposes as
        # model.causal_mask, writable like the reference's, generate.py:154-163) whenever the model is t2i: a caller
        # that folds emb_masks into it by hand and then drives __call__ gets the reference's result.  c2i models
        # never edit the mask in the reference, so they keep the maskless (pure causal) kernels unless use_mask is set.
        self.use_mask = False
        self._force_causal = False  # whole-sequence `is_causal` forward (gpt.py:234) ignores causal_mask
        self._graphs = {}
        self._prof = None
        self.pos_rows = None         # continuous batching (llamagen_amd/serve.py): int32 [MTs*16] per-row positions
        # RMSNorm folded into the consumer GEMMs (5 launches / layer instead of 7).  Pays off (-12 % step time,
        # GPT-L) only with gemm_normpre.hip, which normalises the activation panel in registers while the
        # weight loads are in flight and needs a wave's K range to fit in registers: bf16, d/32 = 8 waves x
        # 3..6 chunks (GPT-B .. GPT-XXL).  Elsewhere the stand-alone 3.4 us norm kernels are faster than the
        # generic NORM prologue (+4 us of exposed VALU work per GEMM).
        kch = self.d // self.kc
        auto = dtype == torch.bfloat16 and kch % 8 == 0 and 3 <= kch // 8 <= 6
        self.use_tile = os.environ.get("LGEN_GEMM_TILE", "1") != "0"
        self.tile_autotune = os.environ.get("LGEN_TILE_AUTOTUNE", "0") == "1"   # opt-in (development): time the tile shapes on the device
        # (round 4) the big-M tile family normalises the B fragments between LDS and the MFMA, whatever the row length: chains of
        # >= 256 rows fuse for every width whose statistics row the kernels take (GPT-3B: d = 3200, 200 partials per row)
        self._fuse_base = auto   # what holds without the tile family (per-row positions: the wqkv of a continuous batcher stays skinny)
        if dtype == torch.bfloat16 and self.MTs >= 16 and self.use_tile and (self.d // 16) % 4 == 0 and self.d // 16 <= L.SSQ_STRIDE:
            auto = True
        env = os.environ.get("LGEN_FUSED_NORM")
        self._fuse_env, self._fuse_auto, self._fuse_forced = env, auto, None   # `fuse_norm` is DERIVED from these + pos_rows (property below)
        self._tile_refused = set()   # GEMM kinds whose table shape the library refused (LGEN_ERR_UNSUPPORTED): served by the skinny kernels, reported as such
        self.tile_override = {}      # kind ("qkv" | "wo" | "w13" | "w2" | "head") -> (mt, nt, kw)
        # tuning hook: LGEN_TILES="qkv=2,4,8;wo=4,1,8;w2=4,1,8" (e.g. fewer, fatter workgroups per GEMM so that the
        # kernels of several in-flight batches share the chip side by side instead of taking turns)
        for item in filter(None, os.environ.get("LGEN_TILES", "").split(";")):
            kind, _, val = item.partition("=")
            if kind.strip() not in ("qkv", "wo", "w13", "w2", "head"):
                raise ValueError(f"LGEN_TILES: unknown GEMM '{kind}'")
            t = tuple(int(v) for v in val.split(","))
            if len(t) != 3:
                raise ValueError(f"LGEN_TILES: '{item}' is not kind=mt,nt,kw")
            self.tile_override[kind.strip()] = t
        # fused-norm GEMM schedule (the `passes` argument of lgen_gemm): kind -> (passes, 0); LGEN_PASSES="qkv=2,0;w13=3,0"
        # big-M tile family (round 4): LGEN_GEMM_TILE=0 keeps the skinny kernels; LGEN_TILE_SHAPES="qkv=4,1,1,3,4,4,4;..." picks shapes
        self.tile_shape_override = {}
        for item in filter(None, os.environ.get("LGEN_TILE_SHAPES", "").split(";")):
            kind, _, val = item.partition("=")
            if kind.strip() not in ("qkv", "wo", "w13", "w2", "head"):
                raise ValueError(f"LGEN_TILE_SHAPES: unknown GEMM '{kind}'")
            t = tuple(int(v) for v in val.split(","))
            if len(t) != 7:
                raise ValueError(f"LGEN_TILE_SHAPES: '{item}' is not kind=wm,wn,mtv,ntv,kb,stages,lw")
            self.tile_shape_override[kind.strip()] = t
        self.pass_override = {}
        for item in filter(None, os.environ.get("LGEN_PASSES", "").split(";")):
            kind, _, val = item.partition("=")
            if kind.strip() not in ("qkv", "w13", "head"):
                raise ValueError(f"LGEN_PASSES: unknown fused-norm GEMM '{kind}'")
            t = tuple(int(v) for v in val.split(","))
            if len(t) != 2 or t[0] < 1:
                raise ValueError(f"LGEN_PASSES: '{item}' is not kind=passes,double_buffer")
            self.pass_override[kind.strip()] = t
        self._pack(model)

    @property
    def pos_rows(self):
        return self._pos_rows

    @pos_rows.setter
    def pos_rows(self, value):
        """Per-row positions (llamagen_amd/serve.py) keep wqkv on the skinny kernels, so the RMSNorm is fused only where the
        register-resident fused-norm GEMMs apply (d / 32 = 8 x 3..6): a wide model (GPT-3B) that fused because its chain is >= 256
        rows would otherwise run the generic norm-prologue kernel, slower than stand-alone norm + the wide ring shapes.
        The fused / unfused choice is the `fuse_norm` property below -- nothing is flipped here (the flip used to stick
        after serving); only the captured graphs of the other mode are dropped, in BOTH directions."""
        before = self.fuse_norm if hasattr(self, "_fuse_auto") else None
        self._pos_rows = value
        if before is not None and self.fuse_norm != before:
            self._graphs = {}

    @property
    def fuse_norm(self) -> bool:
        """RMSNorm inside the consumer GEMMs?  LGEN_FUSED_NORM if set, else an explicit assignment (tests), else: with per-row
        positions only where the register-resident fused-norm kernels apply (`_fuse_base`), otherwise wherever the chain width
        / storage type allows (`_fuse_auto`).  gemm_schedule(), tile_schedule_tested() and the graph key read this."""
        if self._fuse_env is not None:
            return self._fuse_env == "1"
        if self._fuse_forced is not None:
            return self._fuse_forced
        return self._fuse_base if self._pos_rows is not None else self._fuse_auto

    @fuse_norm.setter
    def fuse_norm(self, value):
        self._fuse_forced = None if value is None else bool(value)
        self._graphs = {}

    # ---- weights --------------------------------------------------------------------------
    def _sig(self, model):
        return tuple((p.data_ptr(), p._version) for p in model.parameters())

    def _pack(self, model):
        """Fragment-packed weight copies: built once per parameter set and shared by every engine (lane)
        that serves the same parameters."""
        sig = self._sig(model)
        pw = _PACKED.get(model.tok_embeddings)
        if pw is None or pw.sig != sig or pw.dtype != self.dtype:
            pw = PackedWeights(model, self.dtype, sig)
            _PACKED[model.tok_embeddings] = pw
        self.pw = pw
        self.layers, self.norm_w, self.out_w, self.tok_emb = pw.layers, pw.norm_w, pw.out_w, pw.tok_emb
        if model.model_type == "c2i":
            self.cls_emb = pw.cls_emb
        else:
            self.fc1, self.fc2, self.cap_hidden = pw.fc1, pw.fc2, pw.cap_hidden
        self._wsig = sig
        self._graphs = {}

    def compatible(self, model, max_batch, S8, dtype) -> bool:
        return (self.B2 == max_batch and self.S8 == S8 and self.dtype == dtype
                and self.dev == model.tok_embeddings.weight.device and self._wsig == self._sig(model))

    def reset(self, max_batch: int):
        """What re-running setup_caches means in the reference: fresh caches, position 0.  The KV slabs are
        NOT re-zeroed (3.5 GiB of stores per call at config 2): attention only ever reads slots < kv_len,
        all of which the current call has written (causal), so stale slots are unobservable."""
        self.state.zero_()
        self.use_mask = False
        self.causal_mask.copy_(torch.tril(torch.ones(self.S8, self.S8, dtype=torch.bool, device=self.dev)))

    # ---- tile heuristics --------------------------------------------------------------------
    def _tiles(self, kind: str, N: int, K: int):
        """(mt, nt, kw) for one decode GEMM.  Measured on MI355X (tools/ubench_kernels.py, GPT-L, M = 64):
        every kernel of the chain is latency-bound (~4-6 us), so the shape that wins is the one that puts
        >= ~160 workgroups on the chip with the fewest dependent load rounds per wave: split the batch
        rows over workgroups (mt < MTs) when N alone gives too few tiles (wo / w2: N/16 = 64), group
        n-tiles (nt 2 / 4) only when there are >= 352 / 1024 of them (w1||w3, lm_head)."""
        if kind in self.tile_override:
            return self.tile_override[kind]
        ntiles = N // 16
        kch = K // self.kc
        bf16 = self.dtype == torch.bfloat16
        if self.MTs >= 8 and bf16 and not self.fuse_norm and kch >= 96:
            # wide models (GPT-3B: d 3200, F 8704) at 256 rows: every GEMM is the plain ring kernel and the work is MFMA-shaped
            # (64 GFLOP per layer), so big tiles and FEW K-splitting waves win -- measured (tools/gemm_sweep_wide.py,
            # profiles/r03_wide_sweep.log): qkv (4, 4, 4) 45.0 us against (4, 2, 8) 54.7, wo (4, 4, 4) 18.2 / 26.9, w1||w3 (8, 2, 4)
            # 64.4 / 84.1, w2 (4, 4, 8) 40.1 / 66.6, lm_head (8, 2, 4) 47.5 / 68.4: 4.07 ms of GEMMs per decode step instead of 5.64
            # at 128 rows: qkv (4, 4, 4) 29.5 us against (4, 2, 8) 34.5, wo (4, 2, 4) 12.9 / 16.3, w1||w3 (4, 4, 4) 43.4 / 51.0,
            # w2 (4, 2, 8) 28.8 / 37.1, lm_head (8, 2, 4) 26.3 / 36.9: 2.78 ms per step instead of 3.37
            if self.MTs >= 16:
                mt, nt = {"qkv": (4, 4), "wo": (4, 4), "w13": (8, 2), "w2": (4, 4), "head": (8, 2)}[kind]
            else:
                mt, nt = {"qkv": (4, 4), "wo": (4, 2), "w13": (4, 4), "w2": (4, 2), "head": (8, 2)}[kind]
            while ntiles % nt:
                nt //= 2
            while self.MTs % mt:
                mt //= 2
            if kind == "w13":
                nt = max(nt, 2)
            return mt, nt, (4 if kch <= 128 else 8)
        if self.MTs >= 16 and bf16 and self.fuse_norm and kch // 4 >= 12 and (kind == "w2" or (kind == "wo" and ntiles >= 96)) \
                and ntiles % 2 == 0:
            # RES GEMMs of the fused-norm models at 256 rows: 4 waves x >= 12 chunks run the steady-state ring kernel (counted waits);
            # measured: GPT-L w2 (2, 2, 4) 6.16 us against (4, 1, 8) 7.35 / (2, 2, 8) 6.90; GPT-XXL w2 13.2 / 17.1, wo 5.6 / 8.1
            # (GPT-L's wo, 8 chunks per wave at kw 4, stays (4, 1, 8): 4.64 us)
            return 2, 2, 4
        epi = {"qkv": L.EPI_QKV, "wo": L.EPI_RES, "w2": L.EPI_RES, "w13": L.EPI_SWIGLU, "head": L.EPI_ROWS}[kind]
        fused = self.fuse_norm and kind in ("qkv", "w13", "head")
        if fused and kch % 8 == 0 and 3 <= kch // 8 <= 6:
            # normpre kernel (persistent along N since round 3): every workgroup normalises its own rows of the panel once and
            # walks `_passes` n-groups, so few rows per workgroup and 4 n-tiles per group.  Measured best (tools/gemm_sweep.py,
            # GPT-L, profiles/r03_gemm_sweep.log): qkv (1, 4, 8) at every chain width (64 / 128 / 256 rows: 6.3 / 8.0 / 9.9 us with
            # 1 / 2 / 3 passes), w1||w3 and lm_head (2, 4, 8) (8.5 / 10.7 / 12.6 us and 11.1 / 14.8 / 22.0 us)
            mt = 1 if kind == "qkv" else max(1, self.mt // 2)
            nt = 4
            while ntiles % nt or (kind == "w13" and nt & 1 and nt > 1):
                nt //= 2
            return mt, max(nt, 2 if kind == "w13" else 1), 8
        nt = 4 if ntiles >= 1024 else (2 if (kind == "w13" or ntiles >= 352) else 1)
        while ntiles % nt:
            nt //= 2
        mt = self.mt
        while mt > 1 and (ntiles // nt) * (self.MTs // mt) < 160 and self.MTs % (mt // 2) == 0:
            mt //= 2
        kmax = self.lib.lgen_gemm_max_kw(epi, 1 if fused else 0, mt, nt)
        kw = max(1, min(kmax, 16 if kch >= 64 else 8, kch // 2))
        return mt, nt, kw

    def _tile_shape(self, kind: str):
        """(wm, wn, mtv, ntv, kb, stages, lw) of the big-M tile family (csrc/gemm_tile.hip) for one decode GEMM, or None where the
        skinny kernels stay: chains of >= 256 rows, bf16, RMSNorm fused (the family's norm consumers read the producer's statistics),
        one position for all rows.  Measured on MI355X, GPT-L, 256 rows (tools/gemm_tile_sweep.py, us per launch, skinny -> tile):
        wqkv 10.2 -> 9.8, wo 4.85 -> 4.1, w1||w3 13.1 -> 10.6, w2 7.8 -> 7.7, lm_head 23.2 -> 20.9."""
        if kind in self._tile_refused:
            return None
        if kind in self.tile_shape_override:
            return self.tile_shape_override[kind]
        if not self.use_tile or self.MTs < 16 or self.dtype != torch.bfloat16 or not self.fuse_norm:
            return None
        if kind == "qkv" and (self.pos_rows is not None or self.hd < 16):
            return None
        if (self.d, self.F, self.V) != TABLE_MODEL and self.tile_autotune:
            tuned = self._tuned_shapes()
            if tuned is not None:
                return tuned[kind]
        table = MODEL_TILE_SCHEDULES.get((self.d, self.F, self.V), TILE_SCHEDULES)   # unknown widths: GPT-L's shapes by chain width
        key = tile_schedule_key(self.MTs, table)
        return None if key is None else table[key][kind]

    # ---- on-device shape search (models / chain widths without a measured table) ---------------------------------------------
    def _tile_call(self, kind, w, i, s, x_in, nw):
        """One launch of the tile family for GEMM `kind` of layer i with workgroup shape s; returns the library's status."""
        lib, st, dt, M, mts = self.lib, L.stream(), self.dt, self.B2, self.MTs
        d, F, H, hd, hdp, S8 = self.d, self.F, self.H, self.hd, self.hdp, self.S8
        if kind == "qkv":
            return lib.lgen_gemm_qkv_rope_tile(L.ptr(w["wqkv"]), L.ptr(x_in), L.ptr(self.qbuf), L.ptr(self.k_cache[i]),
                                               L.ptr(self.v_cache[i]), L.ptr(self.freqs_cis), self.state.data_ptr(), M, mts, d, H, hd,
                                               hdp, S8, self.kvs, dt, *s, L.ptr(nw), L.ptr(self.ssq), self.ssq_parts, self.eps, st)
        wp, xp, out, N, K, epi, ssq_out = {
            "wo": (w and w["wo"], self.ap, self.hp, d, d, L.EPI_RES, self.ssq),
            "w13": (w and w["w13"], x_in, self.gp, 2 * F, d, L.EPI_SWIGLU, None),
            "w2": (w and w["w2"], self.gp, self.hp, d, F, L.EPI_RES, self.ssq),
            "head": (self.out_w, x_in, self.logits, self.V, d, L.EPI_ROWS, None)}[kind]
        return lib.lgen_gemm_tile(L.ptr(wp), L.ptr(xp), L.ptr(out), M, mts, N, K, epi, dt, *s, L.ptr(nw),
                                  L.ptr(self.ssq) if nw is not None else 0, self.ssq_parts, self.eps, L.ptr(ssq_out), st)

    def _tuned_shapes(self):
        """{kind: shape | None (skinny kernels)} for this model width and chain width, measured once per process on the device."""
        key = (self.dev.index, self.d, self.F, self.V, self.H, self.MTs, self.pos_rows is not None)
        if key not in _TUNED:
            if torch.cuda.is_current_stream_capturing():
                return None          # (never the first use: prefill and the first decode step run eagerly)
            _TUNED[key] = self._search_tile_shapes()
        return _TUNED[key]

    def _search_tile_shapes(self):
        """Opt-in (LGEN_TILE_AUTOTUNE=1): times every instantiated workgroup shape of the tile family for each decode GEMM as a
        captured pass over ALL layers' weights (nothing stays cache-resident from one launch to the next, as in the decode chain)
        and keeps the fastest TILE shape; the skinny kernel is timed for the report only and is chosen only where the library takes
        no tile shape at all -- the family never depends on a timing (tile and skinny differ in accumulation order).  Runs on the
        engine's own workspaces, which it saves and restores; a few hundred launches, once per process per (model width, chain
        width).  The result only changes speed: every tile shape gives the same bits."""
        keep = {n: getattr(self, n).clone() for n in ("hp", "ap", "gp", "qbuf", "ssq", "logits", "state")}
        pos = int(self.state[0])
        kv_keep = (self.k_cache[:, :, :, pos].clone(), self.v_cache[:, :, :, pos].clone())
        parts_keep = self.ssq_parts
        self.ssq_parts = self.d // 16
        self.hp.normal_(0, 1)
        self.ap.zero_()        # wo / w2 add their product to hp in place: a zero operand keeps the repeated passes finite
        self.gp.zero_()
        self.ssq.fill_(16.0)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()

        def timed(fn):
            """ms of one captured pass of fn (min of 2 replays after one warm replay), or None when fn reports a failure"""
            ok = fn()            # eager: loads the code object, reports unsupported shapes before anything is captured
            if not ok:
                return None
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                g.capture_begin()
                fn()
                g.capture_end()
                g.replay()
                best = float("inf")
                for _ in range(2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1))
            cur.wait_stream(side)
            return best

        head_w = [None] * 4
        nlay = self.layers[:min(len(self.layers), 24)]
        plan = {"qkv": (nlay, "an", TILE_SHAPES_NORM), "w13": (nlay, "fn", TILE_SHAPES_NORM), "head": (head_w, None, TILE_SHAPES_NORM),
                "wo": (nlay, None, TILE_SHAPES_PLAIN), "w2": (nlay, None, TILE_SHAPES_PLAIN)}
        out, report = {}, {}
        saved_override, saved_auto = self.tile_shape_override, self.tile_autotune
        self.tile_autotune = False
        try:
            for kind, (ws, nkey, shapes) in plan.items():
                if kind == "qkv" and self.pos_rows is not None:
                    # continuous batching: wqkv stays on the per-row-position skinny kernel (_tile_shape); timing it here would
                    # append K/V rows at every request's own position, which the save / restore around the search does not cover
                    out[kind] = None
                    continue

                def nw_of(w):
                    return self.norm_w if kind == "head" else (w[nkey] if nkey else None)

                def skinny():
                    self.tile_shape_override = {k: None for k in ("qkv", "wo", "w13", "w2", "head")}   # -> the skinny kernels
                    try:
                        for i, w in enumerate(ws):
                            if kind == "qkv":
                                self.qkv_gemm(i, w, self.hp, nw_of(w))
                            else:
                                self.gemm_kind(kind, w, self.hp, nw_of(w))
                    finally:
                        self.tile_shape_override = saved_override
                    return True

                def tile(s):
                    def run():
                        for i, w in enumerate(ws):
                            if self._tile_call(kind, w, i, s, self.hp, nw_of(w)) != 0:
                                return False
                        return True
                    return run

                best_t, best_s = float("inf"), None
                report[kind] = {"skinny": timed(skinny)}
                for s in shapes:
                    if self.MTs % (s[0] * s[2]) or (kind == "w13" and s[3] % 2):
                        continue
                    t = timed(tile(s))
                    if t is None:
                        continue
                    report[kind][s] = t
                    if t < best_t:
                        best_t, best_s = t, s
                out[kind] = best_s
        finally:
            self.tile_shape_override, self.tile_autotune = saved_override, saved_auto
            torch.cuda.current_stream().synchronize()
            for n, t in keep.items():
                getattr(self, n).copy_(t)
            self.k_cache[:, :, :, pos].copy_(kv_keep[0])
            self.v_cache[:, :, :, pos].copy_(kv_keep[1])
            self.ssq_parts = parts_keep
        self.tile_search_report = report
        return out

    def _passes(self, kind: str, N: int, tiles):
        """(passes, double_buffer) of a fused-norm GEMM: n-groups one workgroup walks with its normalised rows kept in registers
        (gemm_normpre.hip).  One workgroup of these kernels fills a CU, so the (n-group, m-group) units are dealt out as
        ~one workgroup per CU: ceil(units / 256) passes each."""
        if kind in self.pass_override:
            return self.pass_override[kind]
        if not (self.fuse_norm and kind in ("qkv", "w13", "head")):
            return 1, 0
        mt, nt, _ = tiles
        units = (N // 16 // nt) * (self.MTs // max(1, math.gcd(self.MTs, mt)))
        passes = min(64, max(1, -(-units // 256)))   # (the library takes 1..64: a scheduling choice only, results are identical)
        # weights of the next n-group in a second register set: no faster than reloading after the MFMAs (10.74 vs 10.76 us,
        # w1||w3 at 128 rows) and bimodal inside the decode graph (73 / 63 img/s, tools/exp_r3c.py): off
        return passes, 0

    # ---- launches -----------------------------------------------------------------------------
    def _grouped_stats(self):
        """(statistics buffer, partials per row) for a fused-norm consumer of the TILE family: the producer's rows as they are where
        the consumers' fast path takes them, else their 4 lane-group sums, computed here by one small launch (bit-identical)."""
        parts = self.ssq_parts
        fast_path = parts % 16 == 0 and parts <= 128       # the consumers stage such rows in LDS and sum contiguous quarters
        if fast_path or parts % 4 or parts <= 4:           # (parts % 4: lgen_ssq_group4 does not take the row; the consumer refuses it too)
            return self.ssq, parts
        L.check(self.lib.lgen_ssq_group4(L.ptr(self.ssq), L.ptr(self.ssq_g), self.MTs * 16, parts, L.stream()), "ssq_group4")
        return self.ssq_g, 4

    def gemm(self, wp, xp, out, M, mts, N, K, epi, tiles, norm_w=None, ssq_out=None, sched=None, tile=None, kind=None):
        if tile is not None:
            sq, parts = self._grouped_stats() if norm_w is not None else (None, self.ssq_parts)
            rc = self.lib.lgen_gemm_tile(L.ptr(wp), L.ptr(xp), L.ptr(out), M, mts, N, K, epi, self.dt, *tile, L.ptr(norm_w),
                                         L.ptr(sq), parts, self.eps, L.ptr(ssq_out), L.stream())
            if rc != L.ERR_UNSUPPORTED:   # no instantiation / shape does not divide: the skinny kernel below
                L.check(rc, "lgen_gemm_tile")
                return
            if kind is not None:          # ... from now on, and gemm_schedule() / tile_schedule_tested() say so
                self._tile_refused.add(kind)
                self._graphs = {}
        mt, nt, kw = tiles
        if mts % mt:
            mt = math.gcd(mts, mt)
        passes = int(sched[0]) if (sched is not None and norm_w is not None) else 1
        L.check(self.lib.lgen_gemm(L.ptr(wp), L.ptr(xp), L.ptr(out), M, mts, N, K, epi, self.dt, mt, nt, kw,
                                   L.ptr(norm_w), L.ptr(self.ssq) if norm_w is not None else 0, self.ssq_parts, self.eps,
                                   L.ptr(ssq_out), passes, L.stream()), "lgen_gemm")

    def qkv_gemm(self, i, w, x_in, nw):
        """[attention_norm +] wqkv + RoPE + KV append of layer i (gpt.py:214-226): the big-M tile form where it applies, else the
        skinny kernels (per-row positions, narrow chains, other storage types)."""
        lib, st, dt, M, mts = self.lib, L.stream(), self.dt, self.B2, self.MTs
        d, H, hd, hdp, S8 = self.d, self.H, self.hd, self.hdp, self.S8
        rows = self.pos_rows is not None
        pos_ptr = self.pos_rows.data_ptr() if rows else self.state.data_ptr()
        ssq = self.ssq if nw is not None else None
        bq = self._tile_shape("qkv")
        if bq is not None and nw is not None and not rows:
            sq, parts = self._grouped_stats()
            rc = lib.lgen_gemm_qkv_rope_tile(L.ptr(w["wqkv"]), L.ptr(x_in), L.ptr(self.qbuf), L.ptr(self.k_cache[i]),
                                             L.ptr(self.v_cache[i]), L.ptr(self.freqs_cis), pos_ptr, M, mts, d, H, hd, hdp, S8,
                                             self.kvs, dt, *bq, L.ptr(nw), L.ptr(sq), parts, self.eps, st)
            if rc != L.ERR_UNSUPPORTED:
                L.check(rc, "gemm_qkv_rope_tile")
                return
            self._tile_refused.add("qkv")
            self._graphs = {}
        tq = self._tiles("qkv", 3 * d, d)
        sq = self._passes("qkv", 3 * d, tq)
        qkv_fn = lib.lgen_gemm_qkv_rope_rows if rows else lib.lgen_gemm_qkv_rope
        L.check(qkv_fn(L.ptr(w["wqkv"]), L.ptr(x_in), L.ptr(self.qbuf), L.ptr(self.k_cache[i]), L.ptr(self.v_cache[i]),
                       L.ptr(self.freqs_cis), pos_ptr, M, mts, d, H, hd, hdp, S8, self.kvs, dt, tq[0], tq[1], tq[2], L.ptr(nw),
                       L.ptr(ssq), self.ssq_parts, self.eps, sq[0] if nw is not None else 1, st), "gemm_qkv_rope")

    def gemm_kind(self, kind, w, x_in=None, nw=None):
        """One decode GEMM of `kind` ("wo" | "w13" | "w2" | "head") on the decode workspaces with the shapes the decode graph uses
        (also what bench.py's roofline leg and the sweep tools time)."""
        d, F, M, mts = self.d, self.F, self.B2, self.MTs
        fuse = self.fuse_norm
        ssq = self.ssq if fuse else None
        if kind == "wo":
            self.gemm(w["wo"], self.ap, self.hp, M, mts, d, d, L.EPI_RES, self._tiles("wo", d, d), ssq_out=ssq, tile=self._tile_shape("wo"), kind="wo")
        elif kind == "w13":
            t13 = self._tiles("w13", 2 * F, d)
            self.gemm(w["w13"], x_in if x_in is not None else self.hp, self.gp, M, mts, 2 * F, d, L.EPI_SWIGLU, t13, norm_w=nw,
                      sched=self._passes("w13", 2 * F, t13), tile=self._tile_shape("w13") if nw is not None else None, kind="w13")
        elif kind == "w2":
            self.gemm(w["w2"], self.gp, self.hp, M, mts, d, F, L.EPI_RES, self._tiles("w2", d, F), ssq_out=ssq, tile=self._tile_shape("w2"), kind="w2")
        elif kind == "head":
            th = self._tiles("head", self.V, d)
            self.gemm(self.out_w, x_in if x_in is not None else self.hp, self.logits, M, mts, self.V, d, L.EPI_ROWS, th, norm_w=nw,
                      sched=self._passes("head", self.V, th), tile=self._tile_shape("head") if nw is not None else None, kind="head")
        else:
            raise ValueError(kind)

    def tile_schedule_source(self) -> str:
        """Where the tile shapes of gemm_schedule() come from: "override" (LGEN_TILE_SHAPES), "table" (MODEL_TILE_SCHEDULES: measured
        and pinned per registry model), "table (GPT-L shapes by chain width)" (other widths), "search" (LGEN_TILE_AUTOTUNE=1: timed on
        this device at first use), or "none" (chains below 256 rows / other storage types)."""
        if self.tile_shape_override:
            return "override"
        if not self.use_tile or self.MTs < 16 or self.dtype != torch.bfloat16 or not self.fuse_norm:
            return "none"
        if (self.d, self.F, self.V) != TABLE_MODEL and self.tile_autotune:
            return "search"
        return "table" if (self.d, self.F, self.V) in MODEL_TILE_SCHEDULES else "table (GPT-L shapes by chain width)"

    def tile_schedule_tested(self) -> bool:
        """True when every GEMM of the decode graph runs a schedule an end-to-end oracle test names (tests/test_gpu_headline.py):
        the skinny kernels (chains below 256 rows: the round-2/3 tests of every config) or a tested key of this model's tile table."""
        sched = self.gemm_schedule()
        if all(v["family"] == "skinny" for v in sched.values()):
            return not self.tile_shape_override and not self.tile_override and not self.pass_override
        model = (self.d, self.F, self.V)
        table = MODEL_TILE_SCHEDULES.get(model)
        if table is None or self.tile_schedule_source() != "table":
            return False
        names = {"wqkv": "qkv", "wo": "wo", "w13": "w13", "w2": "w2", "lm_head": "head"}
        run = {k: tuple(v.get("shape(wm,wn,mtv,ntv,kb,stages,lw)", ())) for k, v in sched.items()}
        return any(all(run[k] == table[m][names[k]] for k in run) for m in TESTED_MODEL_SCHEDULES.get(model, ()))

    def gemm_schedule(self) -> dict:
        """The kernel family and shapes the decode graph launches per GEMM kind (bench.py prints it; tests pin it)."""
        d, F = self.d, self.F
        dims = {"wqkv": ("qkv", 3 * d, d), "wo": ("wo", d, d), "w13": ("w13", 2 * F, d), "w2": ("w2", d, F), "lm_head": ("head", self.V, d)}
        out = {}
        for name, (kind, N, K) in dims.items():
            ts = self._tile_shape(kind)
            if ts is not None and (self.fuse_norm or kind in ("wo", "w2")):
                out[name] = {"family": "tile", "shape(wm,wn,mtv,ntv,kb,stages,lw)": list(ts)}
            else:
                t = self._tiles(kind, N, K)
                out[name] = {"family": "skinny", "tile(mt,nt,kw)": list(t), "passes": self._passes(kind, N, t)[0]}
        return out

    def _layers_and_logits(self, want_logits: bool = True):
        """L x [attention_norm+wqkv+rope+append | attention | wo+res | ffn_norm+w1,w3+swiglu | w2+res], then
        norm+output: 5 launches per layer (RMSNorm rides in the GEMM prologues, its statistics in the
        epilogues of the GEMMs that produce the residual stream)."""
        lib, st, dt, M, mts = self.lib, L.stream(), self.dt, self.B2, self.MTs
        d, F, H, hd, hdp, S8 = self.d, self.F, self.H, self.hd, self.hdp, self.S8
        # one device scalar (generate(): all rows at the same position) or one position per row (serve.py)
        rows = self.pos_rows is not None
        pos_ptr = self.pos_rows.data_ptr() if rows else self.state.data_ptr()
        attn_fn = lib.lgen_attn_decode_rows if rows else lib.lgen_attn_decode
        fuse = self.fuse_norm
        pm = self._mask()

        nlayers = len(self.layers)
        for i, w in enumerate(self.layers):
            if fuse:
                x_in, nw = self.hp, w["an"]
            else:
                L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(w["an"]), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
                x_in, nw = self.xnp, None
            self.qkv_gemm(i, w, x_in, nw)
            if self._prof is not None:  # bench.py roofline leg: HIP events on the launch stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            L.check(attn_fn(L.ptr(self.qbuf), L.ptr(self.k_cache[i]), L.ptr(self.v_cache[i]), L.ptr(self.ap),
                            pos_ptr, L.ptr(pm), self.T if pm is not None else 0, M, mts, H, hd, hdp, S8, self.kvs, dt, self.attn_variant, st),
                    "attn_decode")
            if self._prof is not None:
                e1.record()
                self._prof["events"].append((e0, e1))
            self.gemm_kind("wo", w)
            self.ssq_parts = d // 16
            if fuse:
                x_in, nw = self.hp, w["fn"]
            else:
                L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(w["fn"]), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
                x_in, nw = self.xnp, None
            self.gemm_kind("w13", w, x_in, nw)
            self.gemm_kind("w2", w)
        if want_logits:
            if fuse:
                x_in, nw = self.hp, self.norm_w
            else:
                L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(self.norm_w), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
                x_in, nw = self.xnp, None  # the next decode step starts there
            self.gemm_kind("head", None, x_in, nw)

    def launches_per_step(self) -> int:
        """Kernel launches of one captured decode step (embed + L layers + norm/lm_head + sampler)."""
        per_layer = 5 if self.fuse_norm else 7
        parts = self.d // 16
        grouped = self.fuse_norm and self._tile_shape("w13") is not None and not ((parts % 16 == 0 and parts <= 128) or parts % 4 or parts <= 4)
        extra = 2 * len(self.layers) + 1 if grouped else 0   # lgen_ssq_group4 in front of wqkv, w1||w3 and lm_head (GPT-3B)
        return 1 + len(self.layers) * per_layer + (1 if self.fuse_norm else 2) + 1 + extra

    def _mask(self):
        if self._force_causal:
            return None
        return self.causal_mask if (self.use_mask or self.model_type == "t2i") else None

    def _embed(self, table, idx, advance: bool = False):
        L.check(self.lib.lgen_embed_pack(L.ptr(table), L.ptr(idx), L.ptr(self.hp), L.ptr(self.ssq) if self.fuse_norm else 0,
                                         L.ptr(self.state) if advance else 0, self.B2, self.MTs, self.d, table.shape[0],
                                         self.dt, L.stream()), "embed_pack")
        self.ssq_parts = self.d // 16

    def _set_residual(self, rows: torch.Tensor):
        """Residual stream <- given rows [B2, d] (t2i prefix tokens from the CaptionEmbedder MLP)."""
        self.hp.copy_(pack_act(rows.contiguous(), self.MTs).view_as(self.hp))
        if self.fuse_norm:
            L.check(self.lib.lgen_ssq_pack(L.ptr(self.hp), L.ptr(self.ssq), self.MTs, self.d, self.dt, L.stream()), "ssq_pack")
        self.ssq_parts = self.d // 16

    def _sample(self, B, sp):
        """CFG mix + top-k + softmax + argmax(p/q) for the token at step = state[1]; q = noise[step]."""
        greedy = 0 if sp["sample_logits"] else 1
        stride = B * self.V if not greedy else 0
        L.check(self.lib.lgen_sample(L.ptr(self.logits), 0 if greedy else L.ptr(self.noise), stride, L.ptr(self.cur_tok),
                                     L.ptr(self.seq), L.ptr(self.state), B, self.V, self.seq.shape[1],
                                     1 if sp["use_cfg"] else 0, float(sp["cfg_scale"]), int(sp["cfg_interval"]),
                                     float(sp["temperature"]), int(sp["top_k"]), float(sp["top_p"]), greedy, self.dt,
                                     L.stream()), "lgen_sample")

    def decode_step(self, B, sp):
        """One KV-cached decode step: advance (pos, step); embed(cur_tok) -> L blocks -> logits -> sample."""
        self._embed(self.tok_emb, self.cur_tok, advance=True)
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
        self.gemm(self.fc1, xp, hid, B2 * T, mts, self.cap_hidden, C, L.EPI_GELU, (mt, 1, 1))
        self.gemm(self.fc2, hid, out, B2 * T, mts, self.d, self.cap_hidden, L.EPI_ROWS, (mt, 1, 1))
        return out[: B2 * T].view(B2, T, self.d)

    # ---- prefix prefill (t2i): all T caption positions through each layer at once -----------------------
    def _run_sequence(self, emb: torch.Tensor):
        """emb [B2, S, d] -> all S positions of all rows through every layer at once (causal, plus the folded
        `emb_masks` when a t2i mask is active): fills KV slots 0..S-1 of every layer and returns the workspace
        whose "hp" holds the final residual stream, rows r = t * B2 + b of a [B2*S]-row packed activation set.
        Same per-row math as feeding the positions one by one (rows are independent except through the
        attention); stand-alone RMSNorm kernels (the fused GEMMs are tuned for M <= 256)."""
        lib, st, dt = self.lib, L.stream(), self.dt
        B2, T, d = emb.shape
        F, H, hd, hdp, S8 = self.F, self.H, self.hd, self.hdp, self.S8
        if B2 != self.B2 or T > S8:
            raise ValueError(f"sequence forward of [{B2}, {T}] rows on caches set up for [{self.B2}, {S8}]")
        R = B2 * T
        mts = _ceil_div(R, 16)
        mts = _ceil_div(mts, 8) * 8 if mts > 4 else (4 if mts == 3 else mts)
        key = (R, mts)
        ws = getattr(self, "_pf_ws", None)
        if ws is None or ws["key"] != key:
            z = lambda *s_: torch.zeros(*s_, dtype=self.dtype, device=self.dev)
            ws = dict(key=key, R=R, mts=mts, hp=z(d // self.kc, mts, 64, self.epl), xn=z(d // self.kc, mts, 64, self.epl),
                      ap=z(d // self.kc, mts, 64, self.epl), gp=z(F // self.kc, mts, 64, self.epl),
                      qkv=z(3 * d // self.kc, mts, 64, self.epl), q=z(mts * 16, H, hdp))
            self._pf_ws = ws
        rows = emb.transpose(0, 1).reshape(R, d).to(self.dtype)  # r = t * B2 + b
        ws["hp"].copy_(pack_act(rows.contiguous(), mts).view_as(ws["hp"]))
        mt = min(mts, 4)
        tile = lambda N, K: (mt, 1, max(1, min(8, (K // self.kc) // 2)))
        pm = self._mask()
        for i, w in enumerate(self.layers):
            L.check(lib.lgen_rmsnorm(L.ptr(ws["hp"]), L.ptr(w["an"]), L.ptr(ws["xn"]), mts, d, self.eps, dt, st), "rmsnorm")
            self.gemm(w["wqkv"], ws["xn"], ws["qkv"], R, mts, 3 * d, d, L.EPI_PACKED, tile(3 * d, d))
            L.check(lib.lgen_rope_append_prefill(L.ptr(ws["qkv"]), L.ptr(ws["q"]), L.ptr(self.k_cache[i]), L.ptr(self.v_cache[i]),
                                                 L.ptr(self.freqs_cis), R, B2, mts, d, H, hd, hdp, S8, self.kvs, 0, dt, st),
                    "rope_append_prefill")
            L.check(lib.lgen_attn_prefill(L.ptr(ws["q"]), L.ptr(self.k_cache[i]), L.ptr(self.v_cache[i]), L.ptr(ws["ap"]), L.ptr(pm),
                                          T, B2, mts, H, hd, hdp, S8, self.kvs, dt, st), "attn_prefill")
            self.gemm(w["wo"], ws["ap"], ws["hp"], R, mts, d, d, L.EPI_RES, tile(d, d))
            L.check(lib.lgen_rmsnorm(L.ptr(ws["hp"]), L.ptr(w["fn"]), L.ptr(ws["xn"]), mts, d, self.eps, dt, st), "rmsnorm")
            self.gemm(w["w13"], ws["xn"], ws["gp"], R, mts, 2 * F, d, L.EPI_SWIGLU, (mt, 2, max(1, min(8, (d // self.kc) // 2))))
            self.gemm(w["w2"], ws["gp"], ws["hp"], R, mts, d, F, L.EPI_RES, tile(d, F))
        return ws

    def _prefill_prefix(self, emb: torch.Tensor):
        """emb [B2, T, d] (CaptionEmbedder output) -> KV slots 0..T-1 of every layer filled, residual stream of
        the LAST prefix position left in the decode workspace (self.hp)."""
        B2, T, _ = emb.shape
        ws = self._run_sequence(emb)
        last = unpack_act(ws["hp"], ws["R"])[(T - 1) * B2:]
        self._set_residual(last)

    def forward_sequence(self, model, idx, cond_idx):
        """Teacher-forced forward over a whole sequence (gpt.py:341-346, 357-368 with `is_causal` attention):
        rows = cls_embedding(cond_idx)[:, :T] ++ tok_embeddings(idx) -> fp32 logits [B, T + n, V] (values carry
        the storage dtype's rounding, gpt.py:368).  The reference runs this shape without KV caches; here the
        K/V of every position land in the engine's slabs (slots 0..T+n-1) as a side effect."""
        if model.model_type == "c2i":
            cond = self.cls_emb[cond_idx.reshape(-1).long()].unsqueeze(1)       # LabelEmbedder, eval (gpt.py:78-83)
        else:
            cond = self.caption_embed(cond_idx)                                    # CaptionEmbedder (gpt.py:110-131)
        cond = cond[:, : self.T]
        emb = torch.cat([cond, self.tok_emb[idx.long()]], dim=1)                  # [B, T + n, d]
        B2, S, d = emb.shape
        self._force_causal = True  # this shape is `is_causal` in the reference (gpt.py:234), no emb_masks
        try:
            ws = self._run_sequence(emb)
        finally:
            self._force_causal = False
        lib, st, dt, mts, R = self.lib, L.stream(), self.dt, ws["mts"], ws["R"]
        L.check(lib.lgen_rmsnorm(L.ptr(ws["hp"]), L.ptr(self.norm_w), L.ptr(ws["xn"]), mts, d, self.eps, dt, st), "rmsnorm")
        lg = torch.empty(mts * 16, self.V, dtype=self.dtype, device=self.dev)
        self.gemm(self.out_w, ws["xn"], lg, R, mts, self.V, d, L.EPI_ROWS, (min(mts, 4), 1, max(1, min(8, (d // self.kc) // 2))))
        return lg[:R].view(S, B2, self.V).transpose(0, 1).float().contiguous()

    def _final_logits(self):
        """norm + output on the decode workspace (gpt.py:367-368)."""
        lib, st, dt, M, mts, d = self.lib, L.stream(), self.dt, self.B2, self.MTs, self.d
        th = self._tiles("head", self.V, d)
        if self.fuse_norm:
            x_in, nw = self.hp, self.norm_w
        else:
            L.check(lib.lgen_rmsnorm(L.ptr(self.hp), L.ptr(self.norm_w), L.ptr(self.xnp), mts, d, self.eps, dt, st), "rmsnorm")
            x_in, nw = self.xnp, None
        self.gemm(self.out_w, x_in, self.logits, M, mts, self.V, d, L.EPI_ROWS, th, norm_w=nw, sched=self._passes("head", self.V, th))

    # ---- Exp(1) noise: what torch.multinomial draws, one [B, V] fp32 exponential_ per sampled token ----
    def _noise_buffer(self, N, B):
        if self.noise is None or self.noise.shape[0] < N or self.noise.shape[1] != B:
            # ones, not empty: the slice of a filler batch (generate.PadBatch) is never drawn, and the sampler divides by it
            self.noise = torch.ones(N, B, self.V, dtype=torch.float32, device=self.dev)
        return self.noise

    def draw_noise(self, N, B, b0, n):
        """The N x [n, V] exponential_ draws of ONE batch of a chain (images b0 .. b0 + n - 1 of B): the same calls, in the same
        order, that this batch's own generate() would make (each [n, V] slice is one contiguous block of the chain's buffer)."""
        buf = self._noise_buffer(N, B)
        for j in range(N):
            buf[j, b0:b0 + n].exponential_(1.0)

    def _noise_begin(self, N, B, noise_seq):
        """noise[i] feeds the sampler of step i.  Injected (tests) or drawn from the device's default
        generator in the reference's order -- N separate [B, V] exponential_ calls -- up front on the
        calling stream (N x ~5 us, < 1 % of a generate()), so that no RNG launch sits inside the captured
        decode step.  (A side stream would overlap it, but HIP maps streams onto 4 hardware queues and a
        lane's side stream then queues behind another lane's whole decode loop.)"""
        self._noise_buffer(N, B)
        if noise_seq is not None:
            self.noise[:N].copy_(noise_seq[:N].to(self.dev))
            return
        for j in range(N):
            self.noise[j].exponential_(1.0)

    # ---- the generate() loop ---------------------------------------------------------------
    def generate(self, model, cond_combined, B, max_new_tokens, emb_masks, sp):
        """prefill + (N-1) decode steps; returns int32 [B, N].  sp: sampling parameter dict."""
        it = self.generate_iter(model, cond_combined, B, max_new_tokens, emb_masks, sp)
        while True:
            try:
                next(it)
            except StopIteration as stop:
                return stop.value

    def generate_iter(self, model, cond_combined, B, max_new_tokens, emb_masks, sp):
        """Generator form of generate(): yields after every enqueued decode step (nothing waits for the GPU),
        so that a scheduler can interleave the host-side submission of several in-flight batches
        (llamagen_amd/pipeline.py); the int32 [B, N] result is the generator's return value."""
        N = max_new_tokens
        T = 1 if model.model_type == "c2i" else cond_combined.shape[1]
        if T + N > self.S8 or T + N - 1 > self.freqs_cis.shape[0]:
            raise IndexError(f"{T} + {N} tokens exceed the caches ({self.S8} slots) / RoPE table ({self.freqs_cis.shape[0]} rows)")
        self._prof = getattr(model, "_prof", None)
        self.state.zero_()
        if emb_masks is not None:  # generate.py:154-163: fold emb_masks into causal_mask, force the diagonal
            cm = self.causal_mask
            cm[:, :, :T] = cm[:, :, :T] & emb_masks.to(self.dev).ne(0).unsqueeze(1)
            cm |= torch.eye(self.S8, dtype=torch.bool, device=self.dev)
            self.use_mask = True
        noise_seq = sp.pop("_noise_seq", None)
        prefilled = sp.pop("_noise_prefilled", False)  # generate.py drew it batch by batch (chains of several batches)
        sampling = bool(sp["sample_logits"])
        if sampling and not prefilled:
            self._noise_begin(N, B, noise_seq)
        # ---- prefill (generate.py:77-86): state = (pos, step) = (T-1, 0) when the first token is sampled
        if model.model_type == "c2i":
            self._embed(self.cls_emb, cond_combined.to(torch.int32).contiguous())
            self._layers_and_logits()
        else:
            emb = self.caption_embed(cond_combined)
            if os.environ.get("LGEN_SEQ_PREFILL") == "1" or T > 128:
                for t in range(T):  # causal prefix, one position at a time through the decode kernels
                    self.state[0] = t
                    self._set_residual(emb[:, t])
                    self._layers_and_logits(want_logits=(t == T - 1))
            else:  # all T positions per layer at once
                self._prefill_prefix(emb)
                self.state[0] = T - 1
                self._final_logits()
        self._sample(B, sp)
        yield 0
        # ---- decode (generate.py:105-123): every step first advances (pos, step)
        key = (B, N, self._mask() is not None, self.fuse_norm, self.use_tile, tuple(sorted(self.tile_shape_override.items())),
               tuple(sorted(self.tile_override.items())),
               tuple(sorted(self.pass_override.items())), sp["use_cfg"], sp["cfg_scale"],
               sp["cfg_interval"], sp["temperature"], sp["top_k"], sp["top_p"], sp["sample_logits"],
               self.noise.data_ptr() if sampling else 0)
        use_graph = os.environ.get("LGEN_NO_GRAPH") is None and N > 3
        i = 1
        if use_graph:
            if i < N:  # first decode step eagerly (also warms every kernel before capture)
                self.decode_step(B, sp)
                i += 1
                yield i - 1
            g = self._graphs.get(key)
            if g is None:  # capture only records the step (on the current stream); nothing executes here
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    self.decode_step(B, sp)
                self._graphs[key] = g
            while i < N:
                g.replay()
                i += 1
                yield i - 1
        else:
            while i < N:
                self.decode_step(B, sp)
                i += 1
                yield i - 1
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
        for p in pos:  # the reference's index_put / freqs_cis[input_pos] raise IndexError (gpt.py:177-185, 356)
            if not (0 <= p < self.S8 and p < self.freqs_cis.shape[0]):
                raise IndexError(f"input_pos {p} outside the caches set up for {self.S8} slots / {self.freqs_cis.shape[0]} RoPE rows")
        for j, p in enumerate(pos):
            self.state[0] = p
            if embs is not None:
                self._set_residual(embs[:, j])
            elif cond_idx is not None:
                self._embed(self.cls_emb, rows)
            else:
                self._embed(self.tok_emb, rows)
            self._layers_and_logits()
            outs.append(self.logits[:nb].float())
        return torch.stack(outs, dim=1)
