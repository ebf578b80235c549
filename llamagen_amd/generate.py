"""generate(): drop-in for autoregressive/models/generate.py:126-176 on the HIP engine.

Same signature, same semantics (CFG batch doubling :129-131/:136-138, cache setup :147-152, t2i
emb_masks folding :154-163, prefill :168-170, N-1 decode steps :172-174, int32 [B, N] result on
cond.device), same consumption of the device's default torch generator (one [B, V] fp32
exponential_ per sampled token, exactly what torch.multinomial draws).  The per-token Python loop of
the reference (decode_n_tokens, :105-123) becomes replays of one captured hipGraph with device-side
position / step counters; sampling (sample(), top_k_top_p_filtering, CFG mix) is one HIP kernel.
"""
from __future__ import annotations

import torch


@torch.no_grad()
def generate(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, **sampling_kwargs):
    """Drop-in generate(): runs the step generator below to completion on the current stream."""
    it = generate_iter(model, cond, max_new_tokens, emb_masks, cfg_scale, cfg_interval, **sampling_kwargs)
    while True:
        try:
            next(it)
        except StopIteration as stop:
            return stop.value


class PadBatch:
    """Filler batch of a decode chain (llamagen_amd/pipeline.py: a last, incomplete chain keeps the lane's chain shape).  Its rows
    are computed and dropped; it draws NO noise, so the device generator ends up where consecutive reference generate() calls
    would leave it."""

    def __init__(self, value):
        self.value = value   # (cond, emb_masks or None): any valid conditioning, e.g. the chain's first batch


def _null_condition(model, cond):
    """The unconditional twin of every row of `cond` (reference generate.py:128-141): class-conditional models reserve class id
    `num_classes` for "no class"; text-conditional models carry a learned [T, C] null caption."""
    if model.model_type == "c2i":
        return torch.full_like(cond, model.num_classes)
    return model.cls_embedding.uncond_embedding.to(cond.dtype).expand_as(cond)


def generate_iter(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, **sampling_kwargs):
    """Same arguments as generate(); a generator that yields after every enqueued decode step and returns
    the int32 [B, N] ids (StopIteration.value).  Call it (and every next()) with the lane's stream current
    and under torch.no_grad()."""
    temperature = sampling_kwargs.pop("temperature", 1.0)
    top_k = sampling_kwargs.pop("top_k", 0)
    top_p = sampling_kwargs.pop("top_p", 1.0)
    sample_logits = sampling_kwargs.pop("sample_logits", True)
    noise_seq = sampling_kwargs.pop("_noise_seq", None)  # test hook: inject the Exp(1) draws [N, B, V]
    # pipeline hook (llamagen_amd/pipeline.py, batches_per_chain > 1): `cond` is the FIRST batch of a chain and these callables
    # yield the conditioning of the batches that share its decode chain.  They are evaluated in the order consecutive reference
    # generate() calls would consume the default generator: labels of batch j, then batch j's N x [n, V] Exp(1) draws, then
    # labels of batch j + 1, ... (sample_c2i_ddp.py:128-140) -- so a seeded run is the same whatever the chain width.
    more_conds = sampling_kwargs.pop("_more_conds", None)
    if sampling_kwargs:
        raise TypeError(f"unexpected sampling arguments: {sorted(sampling_kwargs)}")

    kind = model.model_type
    if kind not in ("c2i", "t2i"):
        raise Exception("please check model type")
    use_cfg = cfg_scale > 1.0
    prefix = 1 if kind == "c2i" else cond.shape[1]          # conditioning positions in front of the image tokens
    prefilled = False
    if more_conds:
        if noise_seq is not None:
            raise NotImplementedError("_more_conds with injected noise: concatenate the batches and their noise yourself")
        n, groups = cond.shape[0], 1 + len(more_conds)
        model.setup_caches(max_batch_size=(2 if use_cfg else 1) * n * groups, max_seq_length=prefix + max_new_tokens,
                           dtype=model.tok_embeddings.weight.dtype)
        parts, mask_parts = [cond], [emb_masks]
        for j in range(groups):
            pad = j > 0 and isinstance(more_conds[j - 1], PadBatch)
            if j > 0:
                item = more_conds[j - 1].value if pad else (more_conds[j - 1]() if callable(more_conds[j - 1]) else more_conds[j - 1])
                c_j, m_j = item if isinstance(item, tuple) else (item, None)   # text-conditional: (caption_embs, emb_masks)
                if c_j.shape != cond.shape:
                    raise ValueError("batches that share a chain must have the same size")
                parts.append(c_j)
                mask_parts.append(m_j if m_j is not None else emb_masks)        # no own mask: the shared one (or none at all)
            if sample_logits and not pad:   # (a filler's slice keeps what it held: ones from the allocation or an earlier batch's draws -- positive, finite, its tokens are dropped)
                model._engine.draw_noise(max_new_tokens, n * groups, j * n, n)
        cond = torch.cat(parts)
        if any(m is not None for m in mask_parts):
            if any(m is None for m in mask_parts):
                raise ValueError("either every batch of a chain carries emb_masks or none does")
            emb_masks = torch.cat(mask_parts)
        prefilled = bool(sample_logits)
    B = cond.shape[0]
    rows = torch.cat([cond, _null_condition(model, cond)]) if use_cfg else cond   # CFG: conditional rows, then their twins
    model.setup_caches(max_batch_size=rows.shape[0], max_seq_length=prefix + max_new_tokens,
                       dtype=model.tok_embeddings.weight.dtype)
    masks = None
    if emb_masks is not None:  # [B, T] validity of the (left-padded) caption tokens; the twins see the same pattern
        assert emb_masks.shape[0] == B
        assert emb_masks.shape[-1] == prefix
        masks = emb_masks.repeat(2, *([1] * (emb_masks.dim() - 1))) if use_cfg else emb_masks

    sp = dict(use_cfg=use_cfg, cfg_scale=float(cfg_scale), cfg_interval=int(cfg_interval),
              temperature=float(temperature), top_k=int(top_k), top_p=float(top_p),
              sample_logits=bool(sample_logits))
    if noise_seq is not None:
        sp["_noise_seq"] = noise_seq
    if prefilled:
        sp["_noise_prefilled"] = True
    return (yield from model._engine.generate_iter(model, rows, B, max_new_tokens, masks, sp))
