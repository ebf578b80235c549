"""Checkpoint plumbing of the reference's sampling drivers, for the drop-in modules.

autoregressive/sample/sample_c2i.py:48-61 (and sample_c2i_ddp.py:80-93, sample_t2i.py) accept four on-disk layouts for
the GPT weights -- an FSDP-consolidated raw state_dict (`--from-fsdp`), DDP `{"model": sd}`, DeepSpeed `{"module": sd}`
and `{"state_dict": sd}` -- and load them with `strict=False` (old checkpoints carry a stray `freqs_cis` buffer, newer
ones do not; KV-cache buffers never match).  The VQ tokenizer is always `checkpoint["model"]`, strict
(sample_c2i.py:32-33).  These helpers give `llamagen_amd` models the same behaviour.
"""
from __future__ import annotations

import argparse
from typing import Any, Dict, Mapping, Union

import torch

_WRAPPERS = ("model", "module", "state_dict")  # sample_c2i.py:51-56, in the reference's order of precedence


def pick_state_dict(checkpoint: Mapping[str, Any], from_fsdp: bool = False) -> Dict[str, torch.Tensor]:
    """The tensor dict inside a reference-format GPT checkpoint object."""
    if from_fsdp:
        return dict(checkpoint)
    for key in _WRAPPERS:
        if key in checkpoint:
            return dict(checkpoint[key])
    raise Exception("please check model weight, maybe add --from-fsdp to run command")  # sample_c2i.py:58


def _read(src: Union[str, Mapping[str, Any]], trusted: bool = False):
    """torch.load of a reference-format checkpoint file.  The reference's trainers save more than tensors
    (autoregressive/train/train_c2i.py:229-236: {"model", "optimizer", "steps", "args": argparse.Namespace}); torch >= 2.6
    unpickles with weights_only=True by default and would reject the Namespace, so it is allow-listed explicitly.  `trusted=True`
    is the reference's own behaviour (sample_c2i.py:49: plain torch.load, full unpickling) for checkpoints that carry other
    Python objects -- only for files you would also hand to the reference."""
    if not (isinstance(src, (str, bytes)) or hasattr(src, "read")):
        return src
    if trusted:
        return torch.load(src, map_location="cpu", weights_only=False)
    with torch.serialization.safe_globals([argparse.Namespace]):
        return torch.load(src, map_location="cpu", weights_only=True)


def load_gpt_checkpoint(model: torch.nn.Module, src: Union[str, Mapping[str, Any]], from_fsdp: bool = False, trusted: bool = False):
    """sample_c2i.py:48-61: sniff the wrapper key, drop buffers that are not parameters here (`freqs_cis`,
    `layers.N.attention.kv_cache.*`, `causal_mask`), `load_state_dict(strict=False)`.  Returns the (missing, unexpected)
    key lists so that a caller can insist on a complete load."""
    sd = pick_state_dict(_read(src, trusted), from_fsdp)
    sd = {k: v for k, v in sd.items() if k != "freqs_cis" and "kv_cache" not in k and k != "causal_mask"}
    res = model.load_state_dict(sd, strict=False)
    if hasattr(model, "_engine"):
        model._engine = None  # packed weight copies are rebuilt lazily from the new parameters
    return list(res.missing_keys), list(res.unexpected_keys)


def load_vq_checkpoint(model: torch.nn.Module, src: Union[str, Mapping[str, Any]], trusted: bool = False):
    """sample_c2i.py:32-33: `vq_model.load_state_dict(checkpoint["model"])` (strict)."""
    ck = _read(src, trusted)
    model.load_state_dict(ck["model"])
    if hasattr(model, "_engine"):
        model._engine = None
