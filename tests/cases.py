"""Parity cases shared by tests/golden/make_golden.py (reference run), the oracle tests and
the GPU tests.  Inputs are regenerated from seeds; only reference OUTPUTS are stored."""
import torch

_TINY = dict(n_layer=2, n_head=4, dim=256, vocab_size=1024, block_size=16, num_classes=10,
             cls_token_num=1, model_type="c2i")
_HD100 = dict(n_layer=2, n_head=8, dim=800, vocab_size=2048, block_size=16, num_classes=10,
              cls_token_num=1, model_type="c2i")
_T2I = dict(n_layer=2, n_head=4, dim=256, vocab_size=1024, block_size=16, cls_token_num=120,
            caption_dim=64, model_type="t2i")
_GPTB = dict(vocab_size=16384, block_size=256, num_classes=1000, cls_token_num=1, model_type="c2i")
_GPTL = dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1, model_type="c2i")   # BASELINE configs[1]


def _c(kwargs, **over):
    d = dict(kwargs=kwargs, dtype="fp32", wseed=1, rseed=11, lin_std=0.05, batch=3, n_new=16,
             cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=100, top_p=1.0, sample_logits=True)
    d.update(over)
    return d


GPT_CASES = {
    # free-running token ids must match bit-exactly in fp32 (SURVEY.md section 7 "hard parts")
    "tiny_cfg4": _c(_TINY),
    "tiny_cfg4_s2": _c(_TINY, wseed=2, rseed=12, batch=5),
    "tiny_nocfg_temp": _c(_TINY, cfg_scale=1.0, top_k=0, temperature=0.8, batch=4),
    "tiny_interval": _c(_TINY, cfg_scale=3.0, cfg_interval=5, top_k=50),
    "tiny_topp": _c(_TINY, cfg_scale=2.0, top_k=0, top_p=0.9),
    "tiny_topk_topp": _c(_TINY, cfg_scale=2.0, top_k=200, top_p=0.8),
    "tiny_greedy": _c(_TINY, sample_logits=False),
    "tiny_b1": _c(_TINY, batch=1, cfg_scale=1.0, top_k=20),
    "hd100_cfg4": _c(_HD100, batch=2, top_k=300),
    "t2i_cfg": _c(_T2I, batch=3, cfg_scale=7.5, top_k=100),
    # bf16: reference tokens are stored and used for TEACHER FORCING; logits compared to tolerance
    "tiny_bf16": _c(_TINY, dtype="bf16", trace_steps=list(range(16))),
    "hd100_bf16": _c(_HD100, dtype="bf16", batch=2, top_k=300, trace_steps=list(range(16))),
    # --precision fp16 (sample_c2i.py:108): same teacher-forced comparison in IEEE half
    "tiny_fp16": _c(_TINY, dtype="fp16", trace_steps=list(range(16))),
    # BASELINE.json configs[0]: LlamaGen-B 256px, single image, cfg 1.0, fp32, top-k 2000
    "gptb_c1": _c(_GPTB, registry="GPT-B", batch=1, n_new=256, cfg_scale=1.0, top_k=2000, lin_std=0.02,
                  trace_steps=[0, 1, 128, 255]),
    "gptb_cfg4": _c(_GPTB, registry="GPT-B", batch=2, n_new=24, cfg_scale=4.0, top_k=2000, lin_std=0.02,
                    trace_steps=[0, 1, 23]),
    # BASELINE.json configs[1] at the reference itself (round 3): LlamaGen-L 384 px (24 layers, d 1024), cfg 4.0, top-k 2000.
    # fp32: 24 free-running tokens, bit-exact.  bf16 (the reference default, sample_c2i.py:108): 8 teacher-forced steps; at depth
    # 24 two correct bf16 evaluations differ by more than the 2-layer bar (GEMM accumulation order; the oracle's own fp32-vs-fp64
    # distance is 3.9-6.7 / 0.58-0.83 ulp), hence the per-case bar.
    "gptl_fp32": _c(_GPTL, registry="GPT-L", batch=2, n_new=24, cfg_scale=4.0, top_k=2000, lin_std=0.02, trace_steps=[0, 1, 23]),
    "gptl_bf16": _c(_GPTL, registry="GPT-L", dtype="bf16", batch=2, n_new=8, cfg_scale=4.0, top_k=2000, lin_std=0.02,
                    trace_steps=[0, 3, 7], ulp_max=8.0, ulp_mean=1.3, agree=0.5),
}


def make_gpt_inputs(case):
    """cond (class ids or caption embeddings) and emb_masks, from the case's rseed."""
    g = torch.Generator().manual_seed(case["rseed"] + 1000)
    kw = case["kwargs"]
    B = case["batch"]
    if kw.get("model_type", "c2i") == "c2i":
        cond = torch.randint(0, kw["num_classes"], (B,), generator=g)
        return cond, None
    T, C = kw["cls_token_num"], kw["caption_dim"]
    emb = torch.randn(B, T, C, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    mask = torch.zeros(B, T, dtype=torch.int64)
    for b in range(B):  # "left padding": valid tokens at the end (sample_t2i.py:92-102)
        mask[b, T - int(lens[b]):] = 1
    emb = emb * mask[:, :, None]
    return emb, mask


VQ_CASES = {
    "vq16_4x4": dict(kind="decode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=3, rseed=21, batch=2, h=4, w=4),
    "vq16_3x5": dict(kind="decode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=3, rseed=22, batch=1, h=3, w=5),
    "vq8_4x4": dict(kind="decode", vq="VQ-8", codebook_size=16384, embed_dim=8, wseed=4, rseed=23, batch=2, h=4, w=4),
    "argmin_6x6": dict(kind="argmin", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=3, rseed=24, batch=2, h=6, w=6),
    "argmin_24x24": dict(kind="argmin", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=5, rseed=25, batch=2, h=24, w=24),
    # BASELINE configs[1] decode shape: one 24 x 24 code grid -> 3 x 384 x 384 (stored as uint8 + every 8th pixel in fp32)
    "vq16_24x24": dict(kind="decode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=9, rseed=30, batch=1, h=24, w=24,
                       store="sampled"),
    # encode(): Encoder convs (stride-2 Downsample) -> quant_conv -> argmin; h, w = IMAGE size
    "enc16_32x32": dict(kind="encode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=6, rseed=26, batch=2, h=32, w=32),
    "enc16_48x32": dict(kind="encode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=6, rseed=27, batch=1, h=48, w=32),
    "enc8_16x16": dict(kind="encode", vq="VQ-8", codebook_size=16384, embed_dim=8, wseed=7, rseed=28, batch=2, h=16, w=16),
    # 256 px batch -> 2 x 16 x 16 = 512 latent vectors: the index-parity gate of VQModel.encode (extract_codes_c2i.py:92-111)
    "enc16_256": dict(kind="encode", vq="VQ-16", codebook_size=16384, embed_dim=8, wseed=8, rseed=29, batch=2, h=256, w=256),
}


def make_vq_inputs(case):
    g = torch.Generator().manual_seed(case["rseed"])
    B, h, w = case["batch"], case["h"], case["w"]
    if case["kind"] == "decode":
        codes = torch.randint(0, case["codebook_size"], (B, h * w), generator=g)
        return dict(codes=codes, shape=[B, case["embed_dim"], h, w])
    if case["kind"] == "encode":
        return dict(x=torch.rand(B, 3, h, w, generator=g) * 2 - 1)
    z = torch.randn(B, case["embed_dim"], h, w, generator=g)
    return dict(z=z)


def noise_stream(rseed: int):
    """The Exp(1) draws the reference consumes on CPU after torch.manual_seed(rseed):
    one [B, V] fp32 exponential_ per sampled token (inside torch.multinomial)."""
    g = torch.Generator().manual_seed(rseed)

    def fn(shape):
        return torch.empty(tuple(shape), dtype=torch.float32).exponential_(1, generator=g)
    return fn


def make_topp_tie_rows(V=16384, rows=6, seed=31):
    """Adversarial top-p inputs: every row has one large tie group (40..400 equal logits at scattered indices) that the nucleus
    boundary cuts through, plus a few larger entries and a low tail.  Returns (logits fp32 [rows, V], top_p list)."""
    g = torch.Generator().manual_seed(seed)
    l = torch.full((rows, V), -6.0) + 0.5 * torch.randn(rows, V, generator=g)
    tops = []
    for r in range(rows):
        perm = torch.randperm(V, generator=g)
        nbig, ntie = 1 + r, 40 * (1 + r % 3) * (1 + r // 3 * 4)
        l[r, perm[:nbig]] = 3.0 + 0.25 * torch.arange(nbig, dtype=torch.float32)
        l[r, perm[nbig:nbig + ntie]] = 1.0
        p = torch.softmax(l[r], -1)
        frac = (0.37, 0.5, 0.81, 0.12, 0.66, 0.95)[r % 6]
        tops.append(float(p[perm[:nbig]].sum() + (frac * ntie) * p[perm[nbig]]))
    return l, tops


# ---- BASELINE configs[2..4] at FULL depth, pinned to the reference (round 4) ------------------------------------------------
# GPT-XXL (48 layers) / GPT-3B (24, head_dim 100) / GPT-XL t2i (36, T = 120, 512 px) in bf16 with two images (four CFG rows):
# prefill, two early teacher-forced decode positions, one LATE position on injected K/V cache contents.  The reference produces
# the CFG-mixed logits of every step in the build container (tests/golden/make_golden.py full_depth); the oracle (CPU) and the HIP
# path (GPU, no CPU model needed) are held to them.  Bars (bf16 ulp of the largest logit; max / mean over the row): what two
# correct bf16 evaluations of a model this deep differ by -- the oracle-vs-reference distance measured when the goldens were
# generated (tests/golden/make_golden.py, round 4, quoted per case below as max / mean for the early steps and the late step)
# sets the bar: max(8 / 1.3, 2 x that), for the oracle AND for the HIP path.
FULL_DEPTH_CASES = {
    "gptxxl_c3": dict(registry="GPT-XXL", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                                      model_type="c2i"), wseed=31, lin_std=0.02, batch=2, cfg_scale=4.0, early=2,
                      late=[575], rseed=41, layers=48, bar_early=(16.0, 2.8), bar_late=(32.0, 5.3)),   # oracle: 7.9 / 1.39, 15.8 / 2.60
    "gpt3b_c4": dict(registry="GPT-3B", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                                    model_type="c2i"), wseed=31, lin_std=0.02, batch=2, cfg_scale=4.0, early=2,
                     late=[420], rseed=42, layers=24, bar_early=(16.0, 3.2), bar_late=(21.0, 4.0)),    # oracle: 7.8 / 1.56, 10.1 / 1.98
    "gptxl_t2i_c5": dict(registry="GPT-XL", kwargs=dict(vocab_size=16384, block_size=1024, cls_token_num=120, caption_dim=2048,
                                                        model_type="t2i"), wseed=31, lin_std=0.02, batch=2, cfg_scale=7.5, early=2,
                         late=[1142], rseed=43, layers=36, bar_early=(10.0, 1.9), bar_late=(23.0, 4.3)),   # oracle: 4.8 / 0.91, 11.3 / 2.11
}


def full_depth_inputs(case):
    """(cond, emb_masks or None, steps) of a FULL_DEPTH case; steps = [(label, tokens [B, 1] or None, input_pos)]: the prefill over
    positions 0..T-1, `early` decode positions T, T+1, ..., then the late positions (their caches come from cache_fill)."""
    kw = case["kwargs"]
    B, T, V = case["batch"], kw["cls_token_num"], kw["vocab_size"]
    g = torch.Generator().manual_seed(case["rseed"])
    if kw["model_type"] == "c2i":
        cond, mask = torch.randint(0, kw["num_classes"], (B,), generator=g), None
    else:
        emb = torch.randn(B, T, kw["caption_dim"], generator=g)
        lens = torch.randint(5, T + 1, (B,), generator=g)
        mask = torch.zeros(B, T, dtype=torch.int64)
        for b in range(B):
            mask[b, T - int(lens[b]):] = 1
        cond = (emb * mask[:, :, None]).to(torch.bfloat16).float()
    steps = [("prefill", None, torch.arange(0, T))]
    for i in range(case["early"]):
        steps.append((f"pos{T + i}", torch.randint(0, V, (B, 1), generator=g), torch.tensor([T + i])))
    for p in case["late"]:
        steps.append((f"late{p}", torch.randint(0, V, (B, 1), generator=g), torch.tensor([p])))
    return cond, mask, steps


def cache_fill(B2, H, S, hd, pos, layer, dt, seed):
    """Pseudo-random storage-dtype K/V for cache slots [0, pos) of one layer: ONE random draw per (seed), rolled along the slot
    axis by a layer-dependent shift (the recipe tests/test_gpu_headline.py::_fill_caches uses)."""
    g = torch.Generator().manual_seed(seed)
    kb = (torch.randn(B2, H, S, hd, generator=g) * 0.6).to(dt)
    vb = (torch.randn(B2, H, S, hd, generator=g) * 0.6).to(dt)
    sh = (layer * 37) % S
    return torch.roll(kb, sh, dims=2)[:, :, :pos], torch.roll(vb, sh, dims=2)[:, :, :pos]
