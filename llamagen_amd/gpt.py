"""Host-side mirror of the reference GPT module for the sampling path.

Mirrors the *interface* of autoregressive/models/gpt.py (reference file:line cited per
symbol) -- `ModelArgs`, `Transformer`, the `GPT_models` registry, state_dict key names,
`setup_caches`, `causal_mask`, `__call__(idx, cond_idx, input_pos) -> (logits, loss)` --
while every FLOP of the inference branches runs in the hand-written HIP library
(`llamagen_amd/csrc`, C ABI in `include/lgen.h`) through `DecodeEngine`.

The nn.Module here is a *parameter holder*: it owns reference-layout tensors so that
`load_state_dict` / `.to(device, dtype)` / checkpoints behave like the reference; the
engine keeps MFMA-fragment-packed copies of them ("data layout in HBM").
Training branches of the reference forward (gpt.py:341-346, 370-380) are out of scope and
raise NotImplementedError.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


def find_multiple(n: int, k: int) -> int:
    """gpt.py:18-21"""
    return n if n % k == 0 else n + k - (n % k)


@dataclass
class ModelArgs:
    """Same fields/defaults as gpt.py:23-50 (dropout knobs are accepted and ignored: the
    engine is inference-only, where every dropout / drop_path is the identity)."""
    dim: int = 4096
    n_layer: int = 32
    n_head: int = 32
    n_kv_head: Optional[int] = None
    multiple_of: int = 256
    ffn_dim_multiplier: Optional[float] = None
    rope_base: float = 10000
    norm_eps: float = 1e-5
    initializer_range: float = 0.02

    token_dropout_p: float = 0.1
    attn_dropout_p: float = 0.0
    resid_dropout_p: float = 0.1
    ffn_dropout_p: float = 0.1
    drop_path_rate: float = 0.0

    num_classes: int = 1000
    caption_dim: int = 2048
    class_dropout_prob: float = 0.1
    model_type: str = "c2i"

    vocab_size: int = 16384
    cls_token_num: int = 1
    block_size: int = 256
    max_batch_size: int = 32
    max_seq_len: int = 2048


class _W(nn.Module):
    """Bias-free weight holder ([out, in] like nn.Linear / [rows, dim] like nn.Embedding)."""

    def __init__(self, rows: int, cols: int, std: Optional[float]):
        super().__init__()
        w = torch.empty(rows, cols)
        if std is None:
            w.zero_()
        else:
            w.normal_(mean=0.0, std=std)
        self.weight = nn.Parameter(w, requires_grad=False)


class _Norm(nn.Module):
    """RMSNorm weight holder, gpt.py:137-148 (eps kept for the engine)."""

    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)


class _LabelEmbedder(nn.Module):
    """gpt.py:56-83: table has num_classes + 1 rows when class_dropout_prob > 0 (CFG row)."""

    def __init__(self, num_classes: int, hidden: int, dropout_prob: float, std: float):
        super().__init__()
        self.embedding_table = _W(num_classes + int(dropout_prob > 0), hidden, std)
        self.num_classes = num_classes


class _MLP(nn.Module):
    """gpt.py:118-131 (fc1 -> gelu(tanh) -> fc2, no bias)."""

    def __init__(self, fin: int, hidden: int, fout: int, std: float):
        super().__init__()
        self.fc1 = _W(hidden, fin, std)
        self.fc2 = _W(fout, hidden, std)


class _CaptionEmbedder(nn.Module):
    """gpt.py:89-115."""

    def __init__(self, in_channels: int, hidden: int, std: float, token_num: int = 120):
        super().__init__()
        self.cap_proj = _MLP(in_channels, hidden, hidden, std)
        self.register_buffer("uncond_embedding", torch.randn(token_num, in_channels) / in_channels ** 0.5)


class _Attention(nn.Module):
    def __init__(self, cfg: ModelArgs):
        super().__init__()
        hd = cfg.dim // cfg.n_head
        n_kv = cfg.n_kv_head if cfg.n_kv_head is not None else cfg.n_head
        if n_kv != cfg.n_head:
            raise NotImplementedError("grouped-query attention is not used by any LlamaGen registry model")
        self.wqkv = _W((cfg.n_head + 2 * n_kv) * hd, cfg.dim, cfg.initializer_range)
        self.wo = _W(cfg.dim, cfg.dim, cfg.initializer_range)
        self.kv_cache = None


class _FeedForward(nn.Module):
    def __init__(self, cfg: ModelArgs):
        super().__init__()
        hidden = int(2 * (4 * cfg.dim) / 3)
        if cfg.ffn_dim_multiplier is not None:
            hidden = int(cfg.ffn_dim_multiplier * hidden)
        hidden = find_multiple(hidden, cfg.multiple_of)
        self.hidden_dim = hidden
        self.w1 = _W(hidden, cfg.dim, cfg.initializer_range)
        self.w3 = _W(hidden, cfg.dim, cfg.initializer_range)
        self.w2 = _W(cfg.dim, hidden, cfg.initializer_range)


class _Block(nn.Module):
    def __init__(self, cfg: ModelArgs):
        super().__init__()
        self.attention = _Attention(cfg)
        self.feed_forward = _FeedForward(cfg)
        self.attention_norm = _Norm(cfg.dim, cfg.norm_eps)
        self.ffn_norm = _Norm(cfg.dim, cfg.norm_eps)


class KVCacheView:
    """What `layer.attention.kv_cache` exposes after setup_caches (gpt.py:170-185): views into
    the engine's [L, B, H, S8, hd_pad] cache slabs."""

    def __init__(self, k: torch.Tensor, v: torch.Tensor):
        self.k_cache = k
        self.v_cache = v


class Transformer(nn.Module):
    """Drop-in for gpt.py:260-386 on the inference path (prefill + KV-cached decode)."""

    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.n_layer = config.n_layer
        self.block_size = config.block_size
        self.num_classes = config.num_classes
        self.model_type = config.model_type
        self.cls_token_num = config.cls_token_num
        std = config.initializer_range
        if self.model_type == "c2i":
            self.cls_embedding = _LabelEmbedder(config.num_classes, config.dim, config.class_dropout_prob, std)
        elif self.model_type == "t2i":
            self.cls_embedding = _CaptionEmbedder(config.caption_dim, config.dim, std)
        else:
            raise Exception("please check model type")  # gpt.py:275
        self.tok_embeddings = _W(config.vocab_size, config.dim, std)
        self.layers = nn.ModuleList([_Block(config) for _ in range(config.n_layer)])
        self.norm = _Norm(config.dim, config.norm_eps)
        self.output = _W(config.vocab_size, config.dim, None)  # zero-init like gpt.py:305
        grid = int(self.block_size ** 0.5)
        assert grid * grid == self.block_size
        self.max_batch_size = -1
        self.max_seq_length = -1
        self.causal_mask = None
        self.freqs_cis = None
        self._engine = None
        self._prof = None

    # ---- reference API -------------------------------------------------------------
    def setup_caches(self, max_batch_size: int, max_seq_length: int, dtype: torch.dtype):
        """gpt.py:316-330.  Allocates (or reuses) the engine state: KV slabs, causal mask,
        2-D RoPE table, packed weights, activation workspaces, captured decode graph."""
        from .engine import DecodeEngine

        dev = self.tok_embeddings.weight.device
        if dev.type != "cuda":
            raise RuntimeError(
                "llamagen_amd.Transformer runs only on an AMD GPU through the HIP library "
                "(no CPU fallback); move the model to 'cuda' first")
        S8 = find_multiple(max_seq_length, 8)
        if self._engine is None or not self._engine.compatible(self, max_batch_size, S8, dtype):
            self._engine = DecodeEngine(self, max_batch_size, S8, dtype)
        eng = self._engine
        eng.reset(max_batch_size)
        self.max_seq_length = S8
        self.max_batch_size = max_batch_size
        self.causal_mask = eng.causal_mask[:max_batch_size]
        self.freqs_cis = eng.freqs_cis
        for i, b in enumerate(self.layers):
            b.attention.kv_cache = KVCacheView(eng.k_cache[i], eng.v_cache[i])

    def forward(self, idx, cond_idx, input_pos=None, targets=None, mask=None, valid=None):
        """gpt.py:332-382 in eval mode.  Prefill (idx None, cond_idx given) and KV-cached decode (cond_idx None)
        return (fp32 logits [B, S, V], None).  With BOTH idx and cond_idx the whole sequence
        cls_embedding(cond_idx) ++ tok_embeddings(idx) goes through causally at once (the reference's
        "training or naive inference" shape, gpt.py:341-346) and `targets` / `valid` produce the cross-entropy of
        gpt.py:373-380 from those logits; no dropout, no backward (train mode raises)."""
        if self.training:
            raise NotImplementedError("train-mode forward (dropout + backward) is outside the sampling hot path")
        if mask is not None:
            raise NotImplementedError("explicit attention masks are not part of the reference's call sites for forward")
        if idx is not None and cond_idx is not None:
            B, S = idx.shape[0], self.cls_token_num + idx.shape[1]
            dtype = self.tok_embeddings.weight.dtype
            if self._engine is None or not self._engine.compatible(self, B, max(find_multiple(S, 8), self.max_seq_length), dtype):
                self.setup_caches(B, S, dtype)
            logits = self._engine.forward_sequence(self, idx, cond_idx)
            loss = None
            if valid is not None:
                loss_all = F.cross_entropy(logits.reshape(-1, logits.size(-1)), targets.reshape(-1), reduction="none")
                valid_all = valid[:, None].repeat(1, targets.shape[1]).reshape(-1)
                loss = (loss_all * valid_all).sum() / max(valid_all.sum(), 1)
            elif targets is not None:
                loss = F.cross_entropy(logits.reshape(-1, logits.size(-1)), targets.reshape(-1))
            return logits, loss
        if targets is not None:
            raise NotImplementedError("targets need the whole-sequence form forward(idx, cond_idx, targets=...)")
        if self._engine is None:
            raise RuntimeError("call setup_caches() before forward (gpt.py:316)")
        return self._engine.forward(self, idx, cond_idx, input_pos), None

    def lane_view(self) -> "Transformer":
        """A second handle on the SAME parameters with its own decode state (KV slabs, workspaces, captured
        graph): what a concurrent in-flight batch on another HIP stream runs on (llamagen_amd/pipeline.py)."""
        import copy
        v = copy.copy(self)  # shallow: shares _parameters / _modules / _buffers
        v._engine = None
        v.causal_mask = None
        v.freqs_cis = None
        v.max_batch_size = v.max_seq_length = -1
        return v

    def get_fsdp_wrap_module_list(self) -> List[nn.Module]:
        return list(self.layers)


# ---- registry, gpt.py:438-467 ----------------------------------------------------------
def _mk(n_layer, n_head, dim):
    def f(**kwargs):
        return Transformer(ModelArgs(n_layer=n_layer, n_head=n_head, dim=dim, **kwargs))
    return f


GPT_7B = _mk(32, 32, 4096)
GPT_3B = _mk(24, 32, 3200)
GPT_1B = _mk(22, 32, 2048)
GPT_XXXL = _mk(48, 40, 2560)
GPT_XXL = _mk(48, 24, 1536)
GPT_XL = _mk(36, 20, 1280)
GPT_L = _mk(24, 16, 1024)
GPT_B = _mk(12, 12, 768)

GPT_models = {
    "GPT-B": GPT_B, "GPT-L": GPT_L, "GPT-XL": GPT_XL, "GPT-XXL": GPT_XXL, "GPT-XXXL": GPT_XXXL,
    "GPT-1B": GPT_1B, "GPT-3B": GPT_3B, "GPT-7B": GPT_7B,
}
