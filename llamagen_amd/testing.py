"""Deterministic synthetic weights for tests, goldens and benchmarks.

There is no network for checkpoints, so every test/bench runs on seeded random weights.
`synth_state_dict` builds a reference-compatible state_dict (same key names / shapes as
autoregressive/models/gpt.py and tokenizer/tokenizer_image/vq_model.py) whose values depend
only on (key name, shape, seed) -- not on module construction order -- so the reference
module (when generating goldens), the oracle and the HIP engine all see identical weights.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0, lin_std: float = 0.02) -> torch.Tensor:
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf = key.split(".")[-1]
    parent = key.split(".")[-2] if "." in key else ""
    if key.endswith("codebook_used"):
        return torch.zeros(shape)
    is_norm = ("norm" in parent) and len(shape) == 1
    if is_norm and leaf == "weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if is_norm and leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    if key == "quantize.embedding.weight":
        return torch.randn(shape, generator=g)
    if key.endswith("uncond_embedding"):
        return torch.randn(shape, generator=g) / math.sqrt(shape[-1])
    if len(shape) == 4:  # conv weight: PyTorch-default-like kaiming uniform
        fan_in = shape[1] * shape[2] * shape[3]
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b * math.sqrt(3.0)
    if len(shape) == 1 and leaf == "bias":
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    return lin_std * torch.randn(shape, generator=g)


def synth_state_dict(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
                     lin_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """named_shapes: iterable of (state_dict key, shape); KV-cache buffers are skipped."""
    out = {}
    for k, shape in named_shapes:
        if "kv_cache" in k or k == "freqs_cis":
            continue
        out[k] = synth_tensor(k, tuple(shape), seed, lin_std)
    return out


def synth_for_module(module: torch.nn.Module, seed: int = 0, lin_std: float = 0.02) -> Dict[str, torch.Tensor]:
    return synth_state_dict(((k, tuple(v.shape)) for k, v in module.state_dict().items()), seed, lin_std)
