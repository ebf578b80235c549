"""CPU oracle for the LlamaGen sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU / numpy, fp32 arithmetic with
explicit storage-dtype rounding points) of the reference algorithm on the path that
`BASELINE.json.north_star` names: the GPT next-token decode loop, the CFG/top-k/top-p
sampler and the VQ-VAE tokenizer (codebook lookup, argmin, conv decoder).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module, and only as the checker / the timed CPU baseline -- never as the product
path.  The product (`llamagen_amd/`) must not import anything from `oracle/`.

Parity status: the reference repository holds NO tests, golden vectors or known-answer
fixtures for this path, so the oracle is pinned against outputs
of the reference itself run in the build container: `tests/golden/make_golden.py`
imports `/root/reference`, runs it on seeded synthetic weights and commits the vectors
under `tests/golden/`; `tests/test_oracle_golden.py` holds this restatement to them
(token ids bit-exact in fp32 mode, logits/pixels to stated tolerances).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------
def _rnd(x: Tensor, dt: torch.dtype) -> Tensor:
    """One storage rounding: fp32 value -> storage dtype -> back to fp32 for arithmetic."""
    if dt == torch.float32:
        return x.float()
    return x.to(dt).float()


def find_multiple(n: int, k: int) -> int:
    """autoregressive/models/gpt.py:18-21"""
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class GPTConfig:
    """Hyper-parameters that matter at inference; autoregressive/models/gpt.py:23-50."""
    dim: int
    n_layer: int
    n_head: int
    vocab_size: int = 16384
    block_size: int = 256
    num_classes: int = 1000
    cls_token_num: int = 1
    caption_dim: int = 2048
    model_type: str = "c2i"
    multiple_of: int = 256
    rope_base: float = 10000.0
    norm_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_head

    @property
    def ffn_dim(self) -> int:
        # autoregressive/models/gpt.py:154-159
        return find_multiple(int(2 * (4 * self.dim) / 3), self.multiple_of)


GPT_SIZES = {  # autoregressive/models/gpt.py:438-461
    "GPT-B": dict(n_layer=12, n_head=12, dim=768),
    "GPT-L": dict(n_layer=24, n_head=16, dim=1024),
    "GPT-XL": dict(n_layer=36, n_head=20, dim=1280),
    "GPT-XXL": dict(n_layer=48, n_head=24, dim=1536),
    "GPT-XXXL": dict(n_layer=48, n_head=40, dim=2560),
    "GPT-1B": dict(n_layer=22, n_head=32, dim=2048),
    "GPT-3B": dict(n_layer=24, n_head=32, dim=3200),
    "GPT-7B": dict(n_layer=32, n_head=32, dim=4096),
}


# ------------------------------------------------------------------------------------
# GPT building blocks
# ------------------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float, dt: torch.dtype) -> Tensor:
    """RMSNorm.forward, autoregressive/models/gpt.py:143-148.

    x (storage dtype values held in fp32) -> fp32 normalise -> ROUND to dt -> * weight
    (dt) -> ROUND to dt.  Two rounding points in bf16 mode."""
    xf = x.float()
    n = xf * torch.rsqrt(torch.mean(xf * xf, dim=-1, keepdim=True) + eps)
    n = _rnd(n, dt)
    return _rnd(n * weight.float(), dt)


def linear(x: Tensor, w: Tensor, dt: torch.dtype) -> Tensor:
    """nn.Linear without bias (gpt.py:161-163,199-200,287): fp32 accumulate, one rounding."""
    return _rnd(x.float() @ w.float().t(), dt)


def precompute_freqs_cis_2d(grid_size: int, n_elem: int, base: float, cls_token_num: int) -> Tensor:
    """2-D RoPE table, autoregressive/models/gpt.py:404-417.

    Returns [cls_token_num + grid^2, n_elem//2, 2] fp32 (cos, sin); the first
    cls_token_num rows are ALL ZERO (so q and k of conditioning tokens become 0)."""
    half_dim = n_elem // 2
    freqs = 1.0 / (base ** (torch.arange(0, half_dim, 2)[: (half_dim // 2)].float() / half_dim))
    t = torch.arange(grid_size)
    freqs = torch.outer(t, freqs)  # [grid, half_dim//2]
    fg = torch.concat(
        [freqs[:, None, :].expand(-1, grid_size, -1), freqs[None, :, :].expand(grid_size, -1, -1)], dim=-1
    )  # [grid, grid, half_dim]
    cache = torch.stack([torch.cos(fg), torch.sin(fg)], dim=-1).flatten(0, 1)
    return torch.cat([torch.zeros(cls_token_num, n_elem // 2, 2), cache])


def apply_rotary_emb(x: Tensor, freqs: Tensor, dt: torch.dtype) -> Tensor:
    """apply_rotary_emb, autoregressive/models/gpt.py:420-430.

    x: [B, S, H, hd] (dt values in fp32); freqs: [S, hd//2, 2].  Interleaved (even, odd)
    pairs, fp32 math, one rounding to dt."""
    B, S, H, hd = x.shape
    xs = x.float().reshape(B, S, H, hd // 2, 2)
    f = freqs.view(1, S, 1, hd // 2, 2)
    o = torch.stack(
        [xs[..., 0] * f[..., 0] - xs[..., 1] * f[..., 1], xs[..., 1] * f[..., 0] + xs[..., 0] * f[..., 1]], dim=-1
    )
    return _rnd(o.flatten(3), dt)


def sdpa_math(q: Tensor, k: Tensor, v: Tensor, mask: Tensor, dt: torch.dtype) -> Tensor:
    """F.scaled_dot_product_attention on the MATH backend (gpt.py:232-236 under
    generate.py:112).  ATen's math path (aten/src/ATen/native/transformers/attention.cpp,
    _scaled_dot_product_attention_math, torch 2.10) upcasts bf16 q/k/v to fp32, scales
    BOTH q and k by sqrt(1/sqrt(hd)), adds a -inf mask, softmaxes in fp32, multiplies by
    v in fp32 and rounds the output once to the input dtype.

    q: [B,H,Sq,hd], k/v: [B,H,Sk,hd], mask: bool broadcastable to [B,H,Sq,Sk]."""
    hd = q.shape[-1]
    sf = math.sqrt(1.0 / math.sqrt(hd))
    qf = q.float() * sf
    kf = k.float() * sf
    att = qf @ kf.transpose(-2, -1)
    att = att.masked_fill(~mask, float("-inf"))
    att = torch.softmax(att, dim=-1)
    return _rnd(att @ v.float(), dt)


def gelu_tanh(x: Tensor) -> Tensor:
    """nn.GELU(approximate='tanh'), gpt.py:124."""
    return F.gelu(x, approximate="tanh")


class GPTOracle:
    """Restatement of Transformer (inference branches only), gpt.py:260-382, with KV cache
    (gpt.py:170-185) and the attention/FFN blocks (gpt.py:188-257).

    `sd` is a reference-compatible state_dict (key names of SURVEY.md section 8b); tensors
    are converted to the storage dtype `dt` (fp32 or bf16) exactly like `.to(dtype)`."""

    def __init__(self, cfg: GPTConfig, sd: Dict[str, Tensor], dt: torch.dtype = torch.float32):
        self.cfg = cfg
        self.dt = dt
        self.p = {k: _rnd(v.detach().float(), dt) for k, v in sd.items() if torch.is_floating_point(v)}
        grid = int(cfg.block_size ** 0.5)
        assert grid * grid == cfg.block_size
        self.grid = grid
        self.freqs = precompute_freqs_cis_2d(grid, cfg.head_dim, cfg.rope_base, cfg.cls_token_num)
        self.k_cache: List[Tensor] = []
        self.v_cache: List[Tensor] = []
        self.causal_mask: Optional[Tensor] = None

    # gpt.py:316-330
    def setup_caches(self, max_batch_size: int, max_seq_length: int):
        cfg = self.cfg
        S8 = find_multiple(max_seq_length, 8)
        self.max_seq_length = S8
        shape = (max_batch_size, cfg.n_head, S8, cfg.head_dim)
        self.k_cache = [torch.zeros(shape) for _ in range(cfg.n_layer)]
        self.v_cache = [torch.zeros(shape) for _ in range(cfg.n_layer)]
        cm = torch.tril(torch.ones(S8, S8, dtype=torch.bool))
        self.causal_mask = cm.unsqueeze(0).repeat(max_batch_size, 1, 1)

    # gpt.py:78-83 / 110-131
    def cond_embed(self, cond: Tensor) -> Tensor:
        cfg, p, dt = self.cfg, self.p, self.dt
        if cfg.model_type == "c2i":
            emb = p["cls_embedding.embedding_table.weight"][cond.long()].unsqueeze(1)
        elif cfg.model_type == "t2i":
            x = _rnd(cond.float(), dt)
            h = linear(x, p["cls_embedding.cap_proj.fc1.weight"], dt)
            h = _rnd(gelu_tanh(h), dt)
            emb = linear(h, p["cls_embedding.cap_proj.fc2.weight"], dt)
        else:
            raise Exception("please check model type")
        return emb[:, : cfg.cls_token_num]

    def forward(self, idx: Optional[Tensor], cond: Optional[Tensor], input_pos: Tensor) -> Tensor:
        """Inference branches of Transformer.forward, gpt.py:347-368.  Returns fp32 logits
        [B, S, V] whose values carry the storage-dtype quantisation (gpt.py:368)."""
        cfg, p, dt = self.cfg, self.p, self.dt
        if cond is not None:
            h = self.cond_embed(cond)
        else:
            h = p["tok_embeddings.weight"][idx.long()]
        B, S, _ = h.shape
        input_pos = input_pos.long()
        mask = self.causal_mask[:B, None, input_pos]  # [B,1,S,S8]
        freqs = self.freqs[input_pos]
        H, hd, d = cfg.n_head, cfg.head_dim, cfg.dim
        for li in range(cfg.n_layer):
            pre = f"layers.{li}."
            # --- attention, gpt.py:207-241
            xn = rms_norm(h, p[pre + "attention_norm.weight"], cfg.norm_eps, dt)
            qkv = linear(xn, p[pre + "attention.wqkv.weight"], dt)
            xq, xk, xv = qkv.split([d, d, d], dim=-1)
            xq = apply_rotary_emb(xq.reshape(B, S, H, hd), freqs, dt).transpose(1, 2)
            xk = apply_rotary_emb(xk.reshape(B, S, H, hd), freqs, dt).transpose(1, 2)
            xv = xv.reshape(B, S, H, hd).transpose(1, 2)
            self.k_cache[li][:B, :, input_pos] = xk  # gpt.py:177-185
            self.v_cache[li][:B, :, input_pos] = xv
            o = sdpa_math(xq, self.k_cache[li][:B], self.v_cache[li][:B], mask, dt)
            o = o.transpose(1, 2).reshape(B, S, d)
            h = _rnd(h + linear(o, p[pre + "attention.wo.weight"], dt), dt)  # gpt.py:255
            # --- feed forward, gpt.py:166-167
            xn = rms_norm(h, p[pre + "ffn_norm.weight"], cfg.norm_eps, dt)
            a1 = linear(xn, p[pre + "feed_forward.w1.weight"], dt)
            a3 = linear(xn, p[pre + "feed_forward.w3.weight"], dt)
            g = _rnd(_rnd(F.silu(a1), dt) * a3, dt)
            h = _rnd(h + linear(g, p[pre + "feed_forward.w2.weight"], dt), dt)  # gpt.py:256
        h = rms_norm(h, p["norm.weight"], cfg.norm_eps, dt)
        return linear(h, p["output.weight"], dt)  # .float() of a dt tensor, gpt.py:368


# ------------------------------------------------------------------------------------
# sampling (generate.py)
# ------------------------------------------------------------------------------------
def top_k_top_p_filtering(logits: Tensor, top_k: int = 0, top_p: float = 1.0) -> Tensor:
    """generate.py:16-54.  logits fp32 [B,V] (modified copy returned).  top-k keeps ties at
    the threshold (strict `<`, generate.py:35)."""
    logits = logits.clone()
    V = logits.size(-1)
    if top_k > 0:
        k = min(max(top_k, 1), V)
        kth = torch.sort(logits, dim=-1, descending=True)[0][..., k - 1, None]
        logits[logits < kth] = -float("inf")
    if top_p < 1.0:
        # stable: ties in ascending index order, which is what torch's GPU sort (a stable radix sort) gives the reference;
        # the CPU sort of the same call is unstable and keeps an arbitrary subset of a tie group -- of the SAME size
        # (tests/golden/topp_ties.npz pins the size against the reference itself)
        sl, si = torch.sort(logits, descending=True, stable=True)
        cp = torch.cumsum(torch.softmax(sl, dim=-1), dim=-1)
        rem = cp > top_p
        rem[..., 1:] = rem[..., :-1].clone()
        rem[..., 0] = False
        rem = rem.scatter(1, si, rem)
        logits[rem] = -float("inf")
    return logits


def default_noise(shape) -> Tensor:
    """The Exp(1) draw torch.multinomial makes internally for num_samples == 1
    (ATen native/Sampling / MultinomialKernel: q = empty_like(p).exponential_(1);
    idx = argmax(p / q)); consumes the default CPU generator exactly like the reference."""
    return torch.empty(shape, dtype=torch.float32).exponential_(1)


def sample(logits_last: Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
           sample_logits: bool = True, noise: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """generate.py:57-66.  logits_last fp32 [B,V]; noise = Exp(1) draws [B,V]."""
    l = logits_last / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        l = top_k_top_p_filtering(l, top_k=top_k, top_p=top_p)
    probs = torch.softmax(l, dim=-1)
    if sample_logits:
        if noise is None:
            noise = default_noise(probs.shape)
        idx = torch.argmax(probs / noise, dim=-1, keepdim=True)
    else:
        idx = torch.argmax(probs, dim=-1, keepdim=True)
    return idx, probs


def cfg_mix(logits: Tensor, cfg_scale: float, cfg_flag: bool = True) -> Tensor:
    """generate.py:79-84 / 94-99: rows [0,B) are conditional, [B,2B) unconditional."""
    if cfg_scale > 1.0:
        cond, uncond = torch.split(logits, len(logits) // 2, dim=0)
        return uncond + (cond - uncond) * cfg_scale if cfg_flag else cond
    return logits


def generate(model: GPTOracle, cond: Tensor, max_new_tokens: int, emb_masks: Optional[Tensor] = None,
             cfg_scale: float = 1.0, cfg_interval: int = -1, temperature: float = 1.0, top_k: int = 0,
             top_p: float = 1.0, sample_logits: bool = True,
             noise_fn: Optional[Callable[[Tuple[int, int]], Tensor]] = None,
             trace: Optional[list] = None, teacher: Optional[Tensor] = None) -> Tensor:
    """generate(), generate.py:126-176 (+ prefill :77-86, decode_one_token :89-102,
    decode_n_tokens :105-123).  Returns int32 [B, max_new_tokens].

    noise_fn(shape) supplies the Exp(1) draw of each sampled token (default: CPU default
    generator, same consumption as torch.multinomial).  `trace` (list) receives the fp32
    pre-sampling logits [B,V] (after CFG mix) of every step; `teacher` [B,N] forces the
    token fed back (teacher forcing) while the sampled ids are still returned."""
    cfg = model.cfg
    noise_fn = noise_fn or default_noise
    if cfg.model_type == "c2i":
        cond_combined = torch.cat([cond, torch.ones_like(cond) * cfg.num_classes]) if cfg_scale > 1.0 else cond
        T = 1
    elif cfg.model_type == "t2i":
        if cfg_scale > 1.0:
            cond_null = torch.zeros_like(cond) + model.p["cls_embedding.uncond_embedding"].to(cond.dtype)
            cond_combined = torch.cat([cond, cond_null])
        else:
            cond_combined = cond
        T = cond.shape[1]
    else:
        raise Exception("please check model type")
    B = cond.shape[0]
    B2 = B * 2 if cfg_scale > 1.0 else B
    model.setup_caches(B2, T + max_new_tokens)
    if emb_masks is not None:  # generate.py:154-163
        assert emb_masks.shape[0] == B and emb_masks.shape[-1] == T
        em = torch.cat([emb_masks, emb_masks]) if cfg_scale > 1.0 else emb_masks
        cm = model.causal_mask
        cm[:, :, :T] = cm[:, :, :T] & (em.unsqueeze(1) != 0)
        eye = torch.eye(cm.size(1), cm.size(2), dtype=torch.bool)
        model.causal_mask = cm | eye

    skw = dict(temperature=temperature, top_k=top_k, top_p=top_p, sample_logits=sample_logits)
    seq = torch.empty((B, max_new_tokens), dtype=torch.int32)
    # prefill
    logits = model.forward(None, cond_combined, torch.arange(0, T))
    logits = cfg_mix(logits, cfg_scale)[:, -1]
    if trace is not None:
        trace.append(logits.clone())
    nxt, _ = sample(logits, noise=noise_fn(logits.shape) if sample_logits else None, **skw)
    seq[:, 0:1] = nxt
    cur = nxt if teacher is None else teacher[:, 0:1].long()
    cfg_flag = True
    for i in range(max_new_tokens - 1):
        if cfg_interval > -1 and i > cfg_interval:
            cfg_flag = False
        x = torch.cat([cur, cur]) if cfg_scale > 1.0 else cur
        logits = model.forward(x.view(-1, 1), None, torch.tensor([T + i]))
        logits = cfg_mix(logits, cfg_scale, cfg_flag)[:, -1]
        if trace is not None:
            trace.append(logits.clone())
        nxt, _ = sample(logits, noise=noise_fn(logits.shape) if sample_logits else None, **skw)
        seq[:, i + 1 : i + 2] = nxt
        cur = nxt if teacher is None else teacher[:, i + 1 : i + 2].long()
    return seq


# ------------------------------------------------------------------------------------
# VQ tokenizer (tokenizer/tokenizer_image/vq_model.py) -- fp32 throughout
# ------------------------------------------------------------------------------------
def l2_normalize(x: Tensor) -> Tensor:
    """F.normalize(p=2, dim=-1, eps=1e-12) as used at vq_model.py:221-223,264."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def get_codebook_entry(codebook: Tensor, indices: Tensor, shape, channel_first: bool = True,
                       l2_norm: bool = True) -> Tensor:
    """VectorQuantizer.get_codebook_entry, vq_model.py:261-276."""
    emb = l2_normalize(codebook) if l2_norm else codebook
    z_q = emb[indices.long().reshape(-1)]
    if shape is not None:
        if channel_first:
            z_q = z_q.reshape(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2).contiguous()
        else:
            z_q = z_q.view(shape)
    return z_q


def codebook_argmin(codebook: Tensor, z: Tensor, l2_norm: bool = True) -> Tensor:
    """VectorQuantizer.forward eval path, vq_model.py:215-232: nearest codebook entry of each
    latent vector.  z: [B, C, H, W] fp32 -> int64 [B*H*W] (first index wins ties, torch.argmin)."""
    zf = z.permute(0, 2, 3, 1).contiguous().view(-1, z.shape[1])
    if l2_norm:
        zf = l2_normalize(zf)
        emb = l2_normalize(codebook)
    else:
        emb = codebook
    d = (zf ** 2).sum(1, keepdim=True) + (emb ** 2).sum(1) - 2 * (zf @ emb.t())
    return torch.argmin(d, dim=1)


def group_norm(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """nn.GroupNorm(32, C, eps=1e-6, affine=True), vq_model.py:359-362 (biased variance)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    var, mean = torch.var_mean(xg, dim=-1, unbiased=False, keepdim=True)
    y = ((xg - mean) * torch.rsqrt(var + eps)).reshape(B, C, H, W)
    return y * w.view(1, C, 1, 1) + b.view(1, C, 1, 1)


def swish(x: Tensor) -> Tensor:
    """nonlinearity, vq_model.py:354-356."""
    return x * torch.sigmoid(x)


def _conv(x, sd, name, padding):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=padding)


def resnet_block(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """ResnetBlock.forward, vq_model.py:299-314 (dropout = identity at eval)."""
    h = swish(group_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]))
    h = _conv(h, sd, pre + "conv1", 1)
    h = swish(group_norm(h, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]))
    h = _conv(h, sd, pre + "conv2", 1)
    if (pre + "nin_shortcut.weight") in sd:
        x = _conv(x, sd, pre + "nin_shortcut", 0)
    return x + h


def attn_block(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """AttnBlock.forward, vq_model.py:327-351 (single head, scale c^-0.5)."""
    h = group_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    q, k, v = (_conv(h, sd, pre + n, 0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = torch.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, sd, pre + "proj_out", 0)


def upsample(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """Upsample.forward, vq_model.py:374-378: nearest 2x then conv3x3."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return _conv(x, sd, pre + "conv", 1)


def vq_decoder(z: Tensor, sd: Dict[str, Tensor], ch_mult=(1, 1, 2, 2, 4), num_res_blocks: int = 2,
               pre: str = "decoder.") -> Tensor:
    """Decoder.forward, vq_model.py:173-194."""
    nres = len(ch_mult)
    h = _conv(z, sd, pre + "conv_in", 1)
    h = resnet_block(h, sd, pre + "mid.0.")
    h = attn_block(h, sd, pre + "mid.1.")
    h = resnet_block(h, sd, pre + "mid.2.")
    for i_level in range(nres):  # conv_blocks are stored in execution order (vq_model.py:146-164)
        has_attn = i_level == 0  # built from i_level == num_resolutions-1 in reversed order
        for i_block in range(num_res_blocks + 1):
            h = resnet_block(h, sd, f"{pre}conv_blocks.{i_level}.res.{i_block}.")
            if has_attn:
                h = attn_block(h, sd, f"{pre}conv_blocks.{i_level}.attn.{i_block}.")
        if i_level != nres - 1:
            h = upsample(h, sd, f"{pre}conv_blocks.{i_level}.upsample.")
    h = swish(group_norm(h, sd[pre + "norm_out.weight"], sd[pre + "norm_out.bias"]))
    return _conv(h, sd, pre + "conv_out", 1)


def downsample(x: Tensor, sd: Dict[str, Tensor], pre: str) -> Tensor:
    """Downsample.forward, vq_model.py:389-393: zero-pad right/bottom by one, conv3x3 stride 2, no padding."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
    return F.conv2d(x, sd[pre + "conv.weight"], sd[pre + "conv.bias"], stride=2, padding=0)


def vq_encoder(x: Tensor, sd: Dict[str, Tensor], ch_mult=(1, 1, 2, 2, 4), num_res_blocks: int = 2,
               pre: str = "encoder.") -> Tensor:
    """Encoder.forward, vq_model.py:105-124."""
    nres = len(ch_mult)
    h = _conv(x, sd, pre + "conv_in", 1)
    for i_level in range(nres):
        for i_block in range(num_res_blocks):
            h = resnet_block(h, sd, f"{pre}conv_blocks.{i_level}.res.{i_block}.")
            if i_level == nres - 1:
                h = attn_block(h, sd, f"{pre}conv_blocks.{i_level}.attn.{i_block}.")
        if i_level != nres - 1:
            h = downsample(h, sd, f"{pre}conv_blocks.{i_level}.downsample.")
    h = resnet_block(h, sd, pre + "mid.0.")
    h = attn_block(h, sd, pre + "mid.1.")
    h = resnet_block(h, sd, pre + "mid.2.")
    h = swish(group_norm(h, sd[pre + "norm_out.weight"], sd[pre + "norm_out.bias"]))
    return _conv(h, sd, pre + "conv_out", 1)


def vq_encode(sd: Dict[str, Tensor], x: Tensor, ch_mult=(1, 1, 2, 2, 4)):
    """VQModel.encode, vq_model.py:41-45 (eval): Encoder -> quant_conv 1x1 -> VectorQuantizer.forward.
    Returns (latent z [B, e_dim, h, w] before quantisation, indices int64 [B*h*w], z_q [B, e_dim, h, w])."""
    sd = {k: v.float() for k, v in sd.items()}
    z = _conv(vq_encoder(x.float(), sd, ch_mult=ch_mult), sd, "quant_conv", 0)
    idx = codebook_argmin(sd["quantize.embedding.weight"], z)
    zq = get_codebook_entry(sd["quantize.embedding.weight"], idx, list(z.shape))
    return z, idx, zq


def vq_decode_code(sd: Dict[str, Tensor], code_b: Tensor, shape, channel_first: bool = True,
                   ch_mult=(1, 1, 2, 2, 4)) -> Tensor:
    """VQModel.decode_code -> decode, vq_model.py:47-55: gather, post_quant_conv 1x1, Decoder."""
    sd = {k: v.float() for k, v in sd.items()}
    quant = get_codebook_entry(sd["quantize.embedding.weight"], code_b, shape, channel_first)
    quant = _conv(quant, sd, "post_quant_conv", 0)
    return vq_decoder(quant, sd, ch_mult=ch_mult)


def to_uint8_hwc(samples: Tensor) -> Tensor:
    """sample_c2i_ddp.py:143: clamp(127.5 x + 128, 0, 255) -> HWC uint8 (truncating cast)."""
    return torch.clamp(127.5 * samples + 128.0, 0, 255).permute(0, 2, 3, 1).to(torch.uint8)
