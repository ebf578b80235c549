"""Parity of the kernels `bench.py` actually times (BASELINE config 2: GPT-L bf16, fused-norm tiles, 32 x 24 x 24
decode_code; 64 rows = one batch per chain, 128 / 256 rows = two / four batches per chain) and of the config 3-5 kernel
shapes against the CPU oracle.

The small-model tests of test_gpu_gpt.py pin the arithmetic; these pin the *instantiations*: the
`gemm_normpre_kernel<BF16, 1, 4, EPI_QKV, 4>` (64 rows) / `<2, 4, EPI_QKV, 4>` (128 rows), `<2, 4, EPI_SWIGLU, 4>`,
`<2, 4, EPI_ROWS, 4>`, `gemm_kernel<BF16, 1, 1, EPI_RES, false, 6>` / `<2, 1, EPI_RES, false, 6>` and
`attn_decode_kernel<BF16, 8, 2, 2>` forms that the headline run replays, at late cache positions too.

Tolerance (bf16 storage, stated here as the prompt asks): CFG-mixed logits within 4 (max) / 0.25 (mean)
bf16 ulp of the largest logit -- the bar of test_forward_teacher_forced_bf16.  A bf16 model is not
invariant to the GEMM accumulation order, so the same distance is also measured
between two legitimate CPU evaluations of the oracle (fp32 vs fp64 accumulation in every nn.Linear); the HIP
path must stay within max(fixed bar, 2 x that self-distance).  Measured on MI355X (gpurun_out/headline_parity.jsonl,
quoted in DESIGN.md): HIP-vs-oracle is 1.0-1.3 x the oracle's own fp32-vs-fp64 distance at every step (GPT-L, 24
layers: 4.2-7.4 ulp max / 0.68-0.97 ulp mean against 3.9-6.7 / 0.58-0.83).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llamagen_oracle as O  # noqa: E402
from tests.util import build_gpt_holder, build_vq_holder, oracle_cfg  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def _L():
    from llamagen_amd import _lib
    return _lib


def _log(name, rec):
    """Keep the measured distances (they back the tolerances quoted in DESIGN.md) when gpurun_out/ exists."""
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "headline_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **rec)) + "\n")


class _Linear64:
    """Context: O.linear with fp64 accumulation (a second legitimate evaluation order of the same model).  cache=True keeps the
    fp64 copy of every weight (a free-running generate() calls each Linear hundreds of times: converting 340 M parameters per step
    is what the time goes to otherwise)."""

    def __init__(self, cache=False):
        self.cache = {} if cache else None

    def _w64(self, w):
        if self.cache is None:
            return w.double()
        k = (w.data_ptr(), tuple(w.shape))
        if k not in self.cache:
            self.cache[k] = w.double()
        return self.cache[k]

    def __enter__(self):
        self.orig = O.linear
        O.linear = lambda x, w, dt: O._rnd((x.double() @ self._w64(w).t()).float(), dt)

    def __exit__(self, *a):
        O.linear = self.orig


def _fill_caches(oracle, eng, pos, hd, seed):
    """Same pseudo-random storage-dtype K/V in slots [0, pos) of every layer of the oracle and the engine."""
    g = torch.Generator().manual_seed(seed)
    B2, H, S8 = oracle.k_cache[0].shape[:3]
    kb = (torch.randn(B2, H, S8, hd, generator=g) * 0.6).to(oracle.dt)
    vb = (torch.randn(B2, H, S8, hd, generator=g) * 0.6).to(oracle.dt)
    kd, vd = kb.to(eng.dev), vb.to(eng.dev)
    for li in range(len(oracle.k_cache)):
        sh = (li * 37) % S8  # a different arrangement per layer, one random draw in total
        oracle.k_cache[li][:, :, :pos] = torch.roll(kb, sh, dims=2)[:, :, :pos].float()
        oracle.v_cache[li][:, :, :pos] = torch.roll(vb, sh, dims=2)[:, :, :pos].float()
        eng.k_cache[li][:, :, :pos, :hd] = torch.roll(kd, sh, dims=2)[:, :, :pos]
        eng.v_cache[li][:, :, :pos, :hd] = torch.roll(vd, sh, dims=2)[:, :, :pos]


def _teacher_forced(case, B, cfg_scale, early, late, cond, emb_masks=None, T=1):
    """Runs prefill + `early` decode positions (+ `late` positions on injected cache contents) through
    Transformer.__call__ on the GPU and through two evaluations of the oracle; returns per-step distances."""
    dev = _dev()
    m, sd = build_gpt_holder(case)
    dt = torch.bfloat16
    m = m.to(device=dev, dtype=dt)
    cfgo = oracle_cfg(case)
    N = case["kwargs"]["block_size"]
    V = case["kwargs"]["vocab_size"]
    B2 = 2 * B
    if cfgo.model_type == "c2i":
        cond_c = torch.cat([cond, torch.ones_like(cond) * cfgo.num_classes])
        cond_dev = cond_c.to(dev)
    else:
        cond_c = torch.cat([cond, torch.zeros_like(cond) + sd["cls_embedding.uncond_embedding"]])
        cond_dev = cond_c.to(dev).to(dt)
    oracles = [O.GPTOracle(cfgo, sd, dt), O.GPTOracle(cfgo, sd, dt)]
    for o in oracles:
        o.setup_caches(B2, T + N)
    m.setup_caches(B2, T + N, dt)
    if emb_masks is not None:  # generate.py:154-163, written through the drop-in API (model.causal_mask)
        em = torch.cat([emb_masks, emb_masks])
        for o in oracles:
            cm = o.causal_mask
            cm[:, :, :T] = cm[:, :, :T] & (em.unsqueeze(1) != 0)
            o.causal_mask = cm | torch.eye(cm.size(1), dtype=torch.bool)
        cmd = m.causal_mask
        cmd[:, :, :T] = cmd[:, :, :T] & (em.to(dev).unsqueeze(1) != 0)
        cmd |= torch.eye(cmd.size(1), dtype=torch.bool, device=dev)
    g = torch.Generator().manual_seed(77)
    steps = []   # (label, idx or None, cond or None, input_pos)
    steps.append(("prefill", None, True, torch.arange(0, T)))
    for i in range(early):
        steps.append((f"pos{T + i}", torch.randint(0, V, (B, 1), generator=g), None, torch.tensor([T + i])))
    for p in late:
        steps.append((f"late{p}", torch.randint(0, V, (B, 1), generator=g), None, torch.tensor([p])))
    out = []
    hd = cfgo.head_dim
    for label, tok, is_cond, ipos in steps:
        if label.startswith("late"):
            p = int(ipos[0])
            for k, o in enumerate(oracles):
                _fill_caches(o, m._engine, p, hd, seed=1000 + p)
        refs = []
        for k, o in enumerate(oracles):
            x = None if tok is None else torch.cat([tok, tok])
            if k == 1:
                with _Linear64():
                    refs.append(o.forward(x, cond_c if is_cond else None, ipos)[:, -1])
            else:
                refs.append(o.forward(x, cond_c if is_cond else None, ipos)[:, -1])
        if is_cond:
            lg, _ = m(None, cond_dev, ipos.to(dev))
        else:
            lg, _ = m(torch.cat([tok, tok]).to(dev), None, ipos.to(dev).to(torch.int))
        got = O.cfg_mix(lg[:, -1].float().cpu(), cfg_scale)
        r0, r1 = O.cfg_mix(refs[0], cfg_scale), O.cfg_mix(refs[1], cfg_scale)
        ulp = r0.abs().max().item() * 2.0 ** -8
        e, s = (got - r0).abs(), (r1 - r0).abs()
        out.append(dict(step=label, ulp=ulp, err_max=e.max().item() / ulp, err_mean=e.mean().item() / ulp,
                        self_max=s.max().item() / ulp, self_mean=s.mean().item() / ulp,
                        argmax_agree=float((got.argmax(-1) == r0.argmax(-1)).float().mean())))
    return out, m


def _check(name, recs):
    for r in recs:
        _log(name, r)
    for r in recs:
        assert r["err_max"] <= max(4.0, 2.0 * r["self_max"]), (name, r)
        assert r["err_mean"] <= max(0.25, 2.0 * r["self_mean"]), (name, r)


def test_config2_gptl_bf16_b64_logits_vs_oracle():
    """BASELINE config 2 instantiations: GPT-L (24 layers, d 1024, 16 heads), bf16, B = 32 -> 64 rows, S8 = 584:
    prefill, positions 1..6, and positions 299 / 574 on injected cache contents."""
    case = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                              model_type="c2i"), wseed=21, lin_std=0.02)
    B = 32
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3))
    recs, m = _teacher_forced(case, B, 4.0, early=6, late=[299, 574], cond=cond)
    e = m._engine
    assert e.fuse_norm and e.MTs == 4 and e.S8 == 584
    assert e._tiles("qkv", 3 * e.d, e.d) == (1, 4, 8) and e._tiles("w13", 2 * e.F, e.d) == (2, 4, 8)  # what bench.py replays
    _check("config2_gptl_b64", recs)


def test_config2_gptl_bf16_two_batches_per_chain_logits_vs_oracle():
    """bench.py's default schedule: two batches of 32 share one decode chain -> 128 rows, MTs = 8, qkv tile (2, 4, 8), RES GEMMs at
    mt = 2: prefill, positions 1..3 and position 299 on injected cache contents, same bar as the 64-row test."""
    case = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                              model_type="c2i"), wseed=21, lin_std=0.02)
    B = 64
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(4))
    recs, m = _teacher_forced(case, B, 4.0, early=3, late=[299], cond=cond)
    e = m._engine
    assert e.fuse_norm and e.MTs == 8 and e.S8 == 584
    assert e._tiles("qkv", 3 * e.d, e.d) == (1, 4, 8) and e._tiles("w13", 2 * e.F, e.d) == (2, 4, 8)
    assert e._passes("qkv", 3 * e.d, (1, 4, 8)) == (2, 0) and e._passes("w13", 2 * e.F, (2, 4, 8)) == (2, 0)
    _check("config2_gptl_b128", recs)


def test_config2_gptl_bf16_four_batches_per_chain_logits_vs_oracle():
    """Round 3 schedule: four batches of 32 share one decode chain -> 256 rows, MTs = 16, the fused-norm GEMMs walk several
    n-groups per workgroup (lgen_gemm_schedule_hint): prefill, positions 1..2 and position 299 on injected cache contents, same
    bar as the 64-row test."""
    case = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                              model_type="c2i"), wseed=21, lin_std=0.02)
    B = 128
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(5))
    recs, m = _teacher_forced(case, B, 4.0, early=2, late=[299], cond=cond)
    e = m._engine
    assert e.fuse_norm and e.MTs == 16 and e.S8 == 584
    sched = e.gemm_schedule()   # round 4: the big-M tile family from 256 rows up
    assert _pinned(16) == {k: tuple(v["shape(wm,wn,mtv,ntv,kb,stages,lw)"]) for k, v in sched.items()}, sched
    _check("config2_gptl_b256", recs)


def test_config2_gptl_bf16_ten_batches_per_chain_logits_vs_oracle():
    """Round 4 schedule (`bench.py --steps 20`: two chains of ten batches): 640 rows, MTs = 40, every decode GEMM on the big-M tile
    family with the 640-row shapes (one round of <= 256 workgroups per launch), persistent decode attention (>= 256 rows): prefill, position 1 and positions 299 / 574
    (the last step of a generate(): the longest key stream of the persistent attention) on injected cache contents, same bar as the
    64-row test; the schedule the bench prints (`roofline_gemm.schedule`) is pinned here."""
    case = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                              model_type="c2i"), wseed=21, lin_std=0.02)
    B = 320
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(6))
    recs, m = _teacher_forced(case, B, 4.0, early=1, late=[299, 574], cond=cond)
    e = m._engine
    assert e.fuse_norm and e.MTs == 40 and e.S8 == 584
    sched = e.gemm_schedule()
    assert all(v["family"] == "tile" for v in sched.values()), sched
    assert _pinned(40) == {k: tuple(v["shape(wm,wn,mtv,ntv,kb,stages,lw)"]) for k, v in sched.items()}, sched
    _check("config2_gptl_b640", recs)


GPTL_CASE = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
                                               model_type="c2i"), wseed=21, lin_std=0.02)


def test_config2_640rows_graph_replay_equals_eager_launches_free_running_bf16():
    """The timed configuration, end to end and free-running: ONE generate() of 320 images (ten batches of 32 = 640 CFG rows, the
    chain shape of `bench.py --steps 20`), GPT-L bf16, all 576 tokens, cfg 4.0, top-k 2000, on the tile GEMMs + persistent
    attention -- once as the product runs it (the captured decode-step hipGraph replayed 574 times, device-side position / step
    counters) and once with every kernel launched eagerly (LGEN_NO_GRAPH=1), from the same Exp(1) draws.  Same kernels, same
    operands: the two token streams must be IDENTICAL (a capture that froze a position, a pointer or a noise offset, or a replay
    that raced the previous step, shows up as a difference somewhere in 320 x 576 tokens)."""
    from llamagen_amd import generate
    dev = _dev()
    m, _ = build_gpt_holder(GPTL_CASE)
    m = m.to(device=dev, dtype=torch.bfloat16)
    B, N, V = 320, 576, 16384
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(8)).to(dev)
    noise = torch.empty(N, B, V, device=dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    for j in range(N):
        noise[j].exponential_(1.0, generator=g)
    kw = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)
    assert os.environ.get("LGEN_NO_GRAPH") is None
    t_graph = generate(m, cond, N, _noise_seq=noise, **kw).cpu()
    e = m._engine
    assert e.MTs == 40 and e.S8 == 584 and len(e._graphs) == 1
    sched = e.gemm_schedule()
    assert _pinned(40) == {k: tuple(v["shape(wm,wn,mtv,ntv,kb,stages,lw)"]) for k, v in sched.items()}, sched
    os.environ["LGEN_NO_GRAPH"] = "1"
    try:
        t_eager = generate(m, cond, N, _noise_seq=noise, **kw).cpu()
    finally:
        del os.environ["LGEN_NO_GRAPH"]
    assert t_graph.shape == (B, N) and int(t_graph.min()) >= 0 and int(t_graph.max()) < V
    same = (t_graph == t_eager)
    assert bool(same.all()), (int((~same).sum()), int((~same).any(1).sum()), int((~same).float().argmax()))
    assert len(torch.unique(t_graph)) > 1000   # a sampled stream, not a constant
    _log("config2_b640_graph_vs_eager", dict(tokens=int(same.numel()), identical=True))


def _rows_fill(eng, kb, vb, rows, pos, hd):
    """K/V slots [0, pos) of every layer of engine `eng` <- rows `rows` of the [R, H, S8, hd] random blocks (layer-rolled as in
    _fill_caches): the same cache contents for the same image whichever chain carries it."""
    S8 = kb.shape[2]
    for li in range(eng.L):
        sh = (li * 37) % S8
        eng.k_cache[li][:, :, :pos, :hd] = torch.roll(kb[rows], sh, dims=2)[:, :, :pos]
        eng.v_cache[li][:, :, :pos, :hd] = torch.roll(vb[rows], sh, dims=2)[:, :, :pos]


def test_config2_bf16_same_batch_alone_and_inside_a_ten_batch_chain():
    """bf16 results depend on the SCHEDULE, within the oracle's own spread: below 256 rows the decode GEMMs are the K-splitting
    skinny kernels (8 waves x K/8, partial sums combined in a fixed order), from 256 rows the tile family (one wave accumulates an
    element over k in order) -- two legitimate fp32 accumulation orders, as different BLAS kernels are for the reference (SURVEY
    8c).  Here the SAME batch of 32 images is evaluated teacher-forced on identical tokens and cache contents (a) alone -- 64 CFG rows,
    what `generate()` of one batch runs -- and (b) as batch 3 of a ten-batch chain -- 640 rows, what `bench.py` times: its
    CFG-mixed logits must agree within the bar every HIP-vs-oracle test uses, max(4 / 0.25 ulp, 2 x the oracle's fp32-vs-fp64
    self-distance); the measured distance goes to gpurun_out/headline_parity.jsonl (INTEGRATION.md quotes it)."""
    dev, dt = _dev(), torch.bfloat16
    m, sd = build_gpt_holder(GPTL_CASE)
    m = m.to(device=dev, dtype=dt)
    cfgo = oracle_cfg(GPTL_CASE)
    n, G, j, N, V, hd = 32, 10, 3, 576, 16384, 64
    g = torch.Generator().manual_seed(91)
    cond = torch.randint(0, 1000, (n * G,), generator=g)
    rows_x = torch.cat([torch.arange(j * n, (j + 1) * n), n * G + torch.arange(j * n, (j + 1) * n)])   # batch j's rows in the chain
    null = torch.ones_like(cond) * cfgo.num_classes
    chain_cond, x_cond = torch.cat([cond, null]), torch.cat([cond, null])[rows_x]
    big, small = m, m.lane_view()
    big.setup_caches(2 * n * G, 1 + N, dt)
    small.setup_caches(2 * n, 1 + N, dt)
    eb, es = big._engine, small._engine
    assert eb.MTs == 40 and es.MTs == 4
    fam_b, fam_s = {v["family"] for v in eb.gemm_schedule().values()}, {v["family"] for v in es.gemm_schedule().values()}
    assert fam_b == {"tile"} and fam_s == {"skinny"}, (fam_b, fam_s)
    oracles = [O.GPTOracle(cfgo, sd, dt), O.GPTOracle(cfgo, sd, dt)]
    for o in oracles:
        o.setup_caches(2 * n, 1 + N)
    S8 = eb.S8
    kb = (torch.randn(2 * n * G, cfgo.n_head, S8, hd, generator=g) * 0.6).to(dt).to(dev)
    vb = (torch.randn(2 * n * G, cfgo.n_head, S8, hd, generator=g) * 0.6).to(dt).to(dev)
    steps = [("prefill", None, torch.arange(0, 1))]
    for i in (1, 2):
        steps.append((f"pos{i}", torch.randint(0, V, (n * G, 1), generator=g), torch.tensor([i])))
    for p in (299, 574):
        steps.append((f"late{p}", torch.randint(0, V, (n * G, 1), generator=g), torch.tensor([p])))
    for label, tok, ipos in steps:
        if label.startswith("late"):
            p = int(ipos[0])
            _rows_fill(eb, kb, vb, torch.arange(2 * n * G, device=dev), p, hd)
            _rows_fill(es, kb, vb, rows_x.to(dev), p, hd)
            kx, vx = kb[rows_x.to(dev)].cpu(), vb[rows_x.to(dev)].cpu()
            for o in oracles:
                for li in range(len(o.k_cache)):
                    sh = (li * 37) % S8
                    o.k_cache[li][:, :, :p] = torch.roll(kx, sh, dims=2)[:, :, :p].float()
                    o.v_cache[li][:, :, :p] = torch.roll(vx, sh, dims=2)[:, :, :p].float()
        if tok is None:
            lb, _ = big(None, chain_cond.to(dev), ipos.to(dev))
            ls, _ = small(None, x_cond.to(dev), ipos.to(dev))
            refs = [oracles[0].forward(None, x_cond, ipos)[:, -1]]
            with _Linear64():
                refs.append(oracles[1].forward(None, x_cond, ipos)[:, -1])
        else:
            t2 = torch.cat([tok, tok])
            lb, _ = big(t2.to(dev), None, ipos.to(dev).to(torch.int))
            ls, _ = small(t2[rows_x].to(dev), None, ipos.to(dev).to(torch.int))
            refs = [oracles[0].forward(t2[rows_x], None, ipos)[:, -1]]
            with _Linear64():
                refs.append(oracles[1].forward(t2[rows_x], None, ipos)[:, -1])
        in_chain = O.cfg_mix(lb[:, -1].float().cpu()[rows_x], 4.0)
        alone = O.cfg_mix(ls[:, -1].float().cpu(), 4.0)
        r0, r1 = O.cfg_mix(refs[0], 4.0), O.cfg_mix(refs[1], 4.0)
        ulp = r0.abs().max().item() * 2.0 ** -8
        d, s_ = (in_chain - alone).abs(), (r1 - r0).abs()
        rec = dict(step=label, ulp=ulp, chain_vs_alone_max=d.max().item() / ulp, chain_vs_alone_mean=d.mean().item() / ulp,
                   self_max=s_.max().item() / ulp, self_mean=s_.mean().item() / ulp,
                   chain_vs_oracle_max=(in_chain - r0).abs().max().item() / ulp, alone_vs_oracle_max=(alone - r0).abs().max().item() / ulp,
                   identical_fraction=float((d == 0).float().mean()),
                   argmax_agree=float((in_chain.argmax(-1) == alone.argmax(-1)).float().mean()))
        _log("config2_same_batch_alone_vs_chain", rec)
        assert rec["chain_vs_alone_max"] <= max(4.0, 2.0 * rec["self_max"]), rec
        assert rec["chain_vs_alone_mean"] <= max(0.25, 2.0 * rec["self_mean"]), rec


def test_bf16_free_running_agreement_report():
    """bf16 free-running token streams cannot be asserted equal against ANY second implementation (the
    reference's own stream changes with the accumulation order of its BLAS: 0 / 16 sequences), so this test REPORTS instead:
    GPT-L (24 layers) bf16, 8 images, cfg 4.0, top-k 2000, all 576 tokens from the same Exp(1) draws -- the HIP path (alone: 16 CFG
    rows on the skinny kernels, AND the same images inside a 640-row tile chain) against the oracle, next to the oracle against its own
    fp64-accumulating evaluation: per image the first diverging step, per 32 steps the fraction of images still on the oracle's
    stream, and at each first divergence the oracle's margin between its two best candidates (log(p/q) gap: a near-tie is what
    flips).  The oracle side (five minutes of CPU) is precomputed: tests/golden/make_free_running_oracle.py ->
    tests/golden/bf16_free_running_oracle.npz.  Written to gpurun_out/r05_bf16_free_running.json (committed under profiles/).
    The only assertions: valid streams, the first token (prefill, no feedback yet) agrees for most images, and the HIP path
    does not fall off the oracle's stream at once where the oracle's own second evaluation stays on it (median first divergence >=
    min(8, 1/4 of that evaluation's)) -- loose on purpose: eight geometric-like samples, a report, not a gate."""
    from llamagen_amd import generate
    from tests.golden.make_free_running_oracle import CASE, B, N, V, KW, inputs
    from tests.util import load_golden
    assert CASE == GPTL_CASE
    gold = load_golden("bf16_free_running_oracle")
    dev, dt = _dev(), torch.bfloat16
    m, _ = build_gpt_holder(GPTL_CASE)
    m = m.to(device=dev, dtype=dt)
    cond, noise = inputs()
    assert abs(float(noise.double().sum()) - float(gold["noise_checksum"])) < 1e-6 * float(gold["noise_checksum"])
    assert np.array_equal(cond.numpy(), gold["cond"])
    ref, ref64, gaps = torch.from_numpy(gold["tokens"]), torch.from_numpy(gold["tokens_fp64_accumulation"]), gold["top2_gap"]
    hip16 = generate(m, cond.to(dev), N, _noise_seq=noise.to(dev), **KW).cpu()
    assert m._engine.MTs == 1
    # the same 8 images as the first 8 of a 320-image chain (640 rows: tile GEMMs + persistent attention); the other images get
    # their own labels and noise
    Bc = 320
    g = torch.Generator().manual_seed(18)
    cond_c = torch.cat([cond, torch.randint(0, 1000, (Bc - B,), generator=g)])
    noise_c = torch.empty(N, Bc, V, device=dev)
    gd = torch.Generator(device=dev).manual_seed(5)
    for jn in range(N):
        noise_c[jn].exponential_(1.0, generator=gd)
    noise_c[:, :B] = noise.to(dev)
    view = m.lane_view()
    hip640 = generate(view, cond_c.to(dev), N, _noise_seq=noise_c, **KW).cpu()[:B]
    assert view._engine.MTs == 40

    def first_div(a, b):
        ne = (a != b)
        return [int(r.float().argmax()) if bool(r.any()) else N for r in ne]

    def survival(fd):
        return [round(sum(1 for f in fd if f > i) / len(fd), 3) for i in range(0, N, 32)]

    rep = {}
    for name, toks in (("hip_16rows_skinny", hip16), ("hip_640rows_tile", hip640), ("oracle_fp64_accumulation", ref64)):
        fd = first_div(toks, ref)
        rep[name] = dict(first_divergence_step=fd, identical_sequences=sum(1 for f in fd if f >= N),
                         on_oracle_stream_every_32_steps=survival(fd),
                         oracle_top2_log_ratio_gap_at_divergence=[round(float(gaps[f, b_]), 5) for b_, f in enumerate(fd) if f < N],
                         token_agreement_overall=round(float((toks == ref).float().mean()), 4))
        assert int(toks.min()) >= 0 and int(toks.max()) < V
    rep["oracle_top2_log_ratio_gap_median_over_all_steps"] = round(float(np.median(gaps)), 5)
    rep["workload"] = "GPT-L 24 layers bf16, 8 images, cfg 4.0, top-k 2000, 576 tokens, same Exp(1) draws; reference = oracle (fp32 accumulation)"
    rep["hip_16rows_vs_hip_640rows_first_divergence_step"] = first_div(hip16, hip640)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump(rep, open(os.path.join(out, "r05_bf16_free_running.json"), "w"), indent=1)
    med = lambda v: sorted(v)[len(v) // 2]
    base = med(rep["oracle_fp64_accumulation"]["first_divergence_step"])
    for name in ("hip_16rows_skinny", "hip_640rows_tile"):
        fd = rep[name]["first_divergence_step"]
        assert sum(1 for f in fd if f >= 1) >= B // 2, (name, fd)
        assert med(fd) >= min(8, base // 4), (name, fd, rep["oracle_fp64_accumulation"]["first_divergence_step"])

def _pinned(min_mts):
    """engine.TILE_SCHEDULES[min_mts] in gemm_schedule()'s naming: what bench.py must print to count as a tested schedule"""
    from llamagen_amd.engine import TESTED_TILE_SCHEDULES, TILE_SCHEDULES
    assert min_mts in TESTED_TILE_SCHEDULES
    t = TILE_SCHEDULES[min_mts]
    return {"wqkv": t["qkv"], "wo": t["wo"], "w13": t["w13"], "w2": t["w2"], "lm_head": t["head"]}


@pytest.mark.parametrize("M", [64, 128, 256])
def test_fused_norm_gemm_passes_bit_identical(M):
    """gemm_normpre.hip walks `passes` n-groups per workgroup with its normalised rows kept in registers (round 3).  The
    arithmetic per output element does not depend on the schedule: qkv (+RoPE+append), w1||w3 (+SwiGLU) and lm_head outputs at
    GPT-L sizes must be BIT-identical for every `passes` (explicit argument since ABI v7), including pass counts that do not divide the n-groups,
    and the one-pass result is held to the oracle."""
    from llamagen_amd.engine import pack_act, pack_weight, precompute_freqs_cis_2d, unpack_act
    from tests.test_gpu_gpt import _close, _rand
    L, dev = _L(), _dev()
    lib = L.lib()
    dt, d, H, hd, F, V, grid, pos = torch.bfloat16, 1024, 16, 64, 2816, 4096, 24, 77
    S8 = O.find_multiple(1 + grid * grid, 8)
    mts = M // 16
    x = _rand((M, d), dt, 51, 1.3)
    nw = (1 + 0.1 * _rand((d,), torch.float32, 52)).to(dt)
    wq, w1, w3, wh = _rand((3 * d, d), dt, 53, 0.03), _rand((F, d), dt, 54, 0.03), _rand((F, d), dt, 55, 0.03), _rand((V, d), dt, 56, 0.03)
    freqs = precompute_freqs_cis_2d(grid, hd, 10000.0, 1)
    xp, nw_d, fr_d = pack_act(x.to(dev), mts), nw.to(dev), freqs.to(dev)
    wqp, whp = pack_weight(wq.to(dev)), pack_weight(wh.to(dev))
    w13 = torch.stack([pack_weight(w1.to(dev)), pack_weight(w3.to(dev))], dim=1).flatten(0, 1).contiguous()
    ssq = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
    L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, L.BF16, L.stream()), "ssq_pack")
    state = torch.tensor([pos, 0], dtype=torch.int32, device=dev)

    def run(tile, passes):
        mt, nt = tile
        kc = torch.zeros(M, H, S8, 64, dtype=dt, device=dev)
        vc = torch.zeros(M, H, S8, 64, dtype=dt, device=dev)
        q = torch.zeros(mts * 16, H, 64, dtype=dt, device=dev)
        gp = torch.zeros(F // 32, mts, 64, 8, dtype=dt, device=dev)
        rows = torch.zeros(mts * 16, V, dtype=dt, device=dev)
        L.check(lib.lgen_gemm_qkv_rope(L.ptr(wqp), L.ptr(xp), L.ptr(q), L.ptr(kc), L.ptr(vc), L.ptr(fr_d), L.ptr(state), M, mts, d, H,
                                       hd, 64, S8, 0, L.BF16, mt, nt, 8, L.ptr(nw_d), L.ptr(ssq), d // 16, 1e-5, passes, L.stream()), "qkv")
        L.check(lib.lgen_gemm(L.ptr(w13), L.ptr(xp), L.ptr(gp), M, mts, 2 * F, d, L.EPI_SWIGLU, L.BF16, mt, nt, 8, L.ptr(nw_d),
                              L.ptr(ssq), d // 16, 1e-5, 0, passes, L.stream()), "w13")
        L.check(lib.lgen_gemm(L.ptr(whp), L.ptr(xp), L.ptr(rows), M, mts, V, d, L.EPI_ROWS, L.BF16, mt, nt, 8, L.ptr(nw_d),
                              L.ptr(ssq), d // 16, 1e-5, 0, passes, L.stream()), "head")
        torch.cuda.synchronize()
        return q, kc[:, :, pos].clone(), vc[:, :, pos].clone(), gp, rows

    # The kernel's own normalised activations, read back through an identity weight matrix (exact: one product per output).
    # They may differ from the oracle's by 1 bf16 ulp in a few elements of a row -- legitimately: the row scale 1/rms comes from
    # a differently ordered fp32 sum, a 1-ulp change of it moves x * scale across a rounding boundary for ALL elements of the row
    # that share a bf16 mantissa (128 distinct ones; tools/diag_m256.py: 12 elements of row 238 at 256 rows) -- so the GEMM +
    # epilogue below are held to the oracle's nn.Linear on the activations the kernel really used.
    eyep = pack_weight(torch.eye(d, dtype=dt).to(dev))
    xrows = torch.zeros(mts * 16, d, dtype=dt, device=dev)
    L.check(lib.lgen_gemm(L.ptr(eyep), L.ptr(xp), L.ptr(xrows), M, mts, d, d, L.EPI_ROWS, L.BF16, 2, 4, 8, L.ptr(nw_d), L.ptr(ssq),
                          d // 16, 1e-5, 0, 1, L.stream()), "norm + identity")
    xn_ref = O.rms_norm(x.float(), nw, 1e-5, dt)
    xn = xrows[:M].float().cpu()
    dx = (xn - xn_ref).abs()
    # a flipped element moved by one bf16 ulp of the INTERMEDIATE rnd(x * scale); times a norm weight < 1 that can be two ulps of
    # the (smaller-binade) result
    assert (dx > 0).float().mean().item() < 1e-3 and (dx <= torch.maximum(xn.abs(), xn_ref.abs()) * 2.0 ** -6 * 1.01).all(), \
        ((dx > 0).sum().item(), dx.max().item())
    ref_rows = O.linear(xn, wh.float(), dt)
    a1, a3 = O.linear(xn, w1.float(), dt), O.linear(xn, w3.float(), dt)
    ref_gp = O._rnd(O._rnd(torch.nn.functional.silu(a1), dt) * a3, dt)
    ref_v = O.linear(xn, wq.float(), dt)[:, 2 * d:].reshape(M, H, hd)
    first = None
    for tile in [(2, 4), (1, 4), (2, 2), (4, 2)]:  # qkv (4, 2) has no multi-pass form (register budget): falls back to one pass
        if tile[0] > mts:
            continue
        base = run(tile, 1)
        for passes in (2, 3, 5, 7, 64):
            got = run(tile, passes)
            for name, a, b in zip(("q", "k", "v", "swiglu", "logits"), base, got):
                assert torch.equal(a, b), (tile, passes, name, (a.float() - b.float()).abs().max().item())
        # the K split (8 waves x 4 chunks, summed wave 0..7) is the same for every workgroup shape: tiles agree bit for bit too
        if first is None:
            first = base
        for name, a, b in zip(("q", "k", "v", "swiglu", "logits"), first, base):
            assert torch.equal(a, b), (tile, "vs first tile", name, (a.float() - b.float()).abs().max().item())
        q, k, v, gp, rows = base
        assert torch.isfinite(rows[:M].float()).all() and torch.isfinite(gp.float()).all()
        _close(rows[:M], ref_rows, dt, f"norm+lm_head {tile}", frac_ulp1=0.05)
        _close(unpack_act(gp, M), ref_gp, dt, f"norm+swiglu {tile}", frac_ulp1=0.08, ulps=3)
        _close(v, ref_v, dt, f"v row {tile}", frac_ulp1=0.05)


TILE_NORM = [(4, 1, 1, 3, 4, 4, 4), (4, 1, 1, 6, 4, 3, 4), (4, 1, 2, 8, 2, 4, 4), (4, 1, 1, 4, 4, 4, 4), (4, 1, 1, 6, 2, 4, 4),
             (4, 1, 1, 8, 2, 4, 4), (4, 1, 2, 4, 2, 4, 4), (4, 1, 2, 6, 2, 4, 4), (4, 1, 1, 2, 4, 4, 4), (4, 1, 1, 3, 4, 4, 0),
             (8, 1, 1, 8, 2, 4, 4), (8, 1, 1, 6, 2, 4, 4), (8, 1, 1, 4, 2, 4, 4), (4, 1, 2, 10, 2, 4, 4)]
TILE_PLAIN = [(2, 2, 1, 1, 4, 4, 4), (2, 2, 1, 2, 4, 4, 4), (2, 2, 2, 1, 4, 4, 4), (2, 2, 2, 2, 4, 4, 4), (2, 2, 2, 2, 2, 4, 4),
              (4, 1, 1, 2, 4, 4, 4), (2, 2, 1, 1, 4, 4, 0), (2, 2, 4, 1, 4, 4, 4), (2, 2, 4, 2, 2, 4, 4)]


TILE_DIMS = {"L": (1024, 16, 64, 2816), "XXL": (1536, 24, 64, 4096), "3B": (3200, 32, 100, 8704)}   # d, heads, head_dim, F


@pytest.mark.parametrize("M,width", [(128, "L"), (256, "L"), (512, "L"), (640, "L"), (256, "XXL"), (256, "3B")])
def test_tile_gemm_family_vs_oracle_and_bit_identical_across_shapes(M, width):
    """Big-M tile family (csrc/gemm_tile.hip, round 4) at GPT-L sizes: every instantiated workgroup shape of lgen_gemm_tile /
    lgen_gemm_qkv_rope_tile -- loader waves or not, any tile, any ring depth -- accumulates an output element in ONE wave over k
    in order, so ALL shapes must agree BIT for bit; the first is held to the oracle: fused RMSNorm (scales bit-identical to the
    skinny kernels': same statistics, same summation order) + wqkv + RoPE + KV append, + w1||w3 + SwiGLU, + lm_head rows, and the
    plain form wo / w2 + residual + the next norm's statistics.  (F = 2816: K = 88 chunks for w2; N = 5632 does not divide every
    tile width: the ragged last n-group is exercised.)  Widths XXL and 3B: the same at GPT-XXL / GPT-3B sizes -- 3B has head_dim 100
    (q / K / V rows padded to 128, q|k|v sections that are no multiple of a 16-column tile) and 200 partial sums of squares per
    row: too many for the LDS staging of the larger shapes, which read them from global memory instead (same order, same bits)."""
    from llamagen_amd.engine import pack_act, pack_weight, precompute_freqs_cis_2d, unpack_act
    from tests.test_gpu_gpt import _close, _rand
    L, dev = _L(), _dev()
    lib = L.lib()
    dt, V, grid, pos = torch.bfloat16, 2048, 24, 77
    d, H, hd, F = TILE_DIMS[width]
    hdp = 64 if hd <= 64 else 128
    S8 = O.find_multiple(1 + grid * grid, 8)
    mts = M // 16
    x = _rand((M, d), dt, 61, 1.3)
    nw = (1 + 0.1 * _rand((d,), torch.float32, 62)).to(dt)
    wq, w1, w3, wh = _rand((3 * d, d), dt, 63, 0.03), _rand((F, d), dt, 64, 0.03), _rand((F, d), dt, 65, 0.03), _rand((V, d), dt, 66, 0.03)
    wo, w2 = _rand((d, d), dt, 67, 0.03), _rand((d, F), dt, 68, 0.02)
    a_in, g_in, h0 = _rand((M, d), dt, 69, 1.0), _rand((M, F), dt, 70, 0.7), _rand((M, d), dt, 71, 1.5)
    freqs = precompute_freqs_cis_2d(grid, hd, 10000.0, 1)
    xp, nw_d, fr_d = pack_act(x.to(dev), mts), nw.to(dev), freqs.to(dev)
    ap, gpin, hp0 = pack_act(a_in.to(dev), mts), pack_act(g_in.to(dev), mts), pack_act(h0.to(dev), mts)
    wqp, whp, wop, w2p = (pack_weight(t.to(dev)) for t in (wq, wh, wo, w2))
    w13 = torch.stack([pack_weight(w1.to(dev)), pack_weight(w3.to(dev))], dim=1).flatten(0, 1).contiguous()
    ssq = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
    L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, L.BF16, L.stream()), "ssq_pack")
    state = torch.tensor([pos, 0], dtype=torch.int32, device=dev)

    def run_norm(s, ssq=ssq, parts=d // 16):
        kc = torch.zeros(M, H, S8, hdp, dtype=dt, device=dev)
        vc = torch.zeros(M, H, S8, hdp, dtype=dt, device=dev)
        q = torch.zeros(mts * 16, H, hdp, dtype=dt, device=dev)
        gp = torch.zeros(F // 32, mts, 64, 8, dtype=dt, device=dev)
        rows = torch.zeros(mts * 16, V, dtype=dt, device=dev)
        rcs = [lib.lgen_gemm_qkv_rope_tile(L.ptr(wqp), L.ptr(xp), L.ptr(q), L.ptr(kc), L.ptr(vc), L.ptr(fr_d), L.ptr(state), M, mts, d,
                                           H, hd, hdp, S8, 0, L.BF16, *s, L.ptr(nw_d), L.ptr(ssq), parts, 1e-5, L.stream())]
        if s[3] % 2 == 0:
            rcs.append(lib.lgen_gemm_tile(L.ptr(w13), L.ptr(xp), L.ptr(gp), M, mts, 2 * F, d, L.EPI_SWIGLU, L.BF16, *s, L.ptr(nw_d),
                                          L.ptr(ssq), parts, 1e-5, 0, L.stream()))
        else:
            gp = None
        rcs.append(lib.lgen_gemm_tile(L.ptr(whp), L.ptr(xp), L.ptr(rows), M, mts, V, d, L.EPI_ROWS, L.BF16, *s, L.ptr(nw_d),
                                      L.ptr(ssq), parts, 1e-5, 0, L.stream()))
        torch.cuda.synchronize()
        if any(rc == L.ERR_UNSUPPORTED for rc in rcs):
            return None
        assert all(rc == 0 for rc in rcs), (s, rcs)
        kw = kc[:, :, pos].clone()
        vw = vc[:, :, pos].clone()
        kc[:, :, pos] = 0
        vc[:, :, pos] = 0
        assert not kc.any() and not vc.any(), s     # nothing but slot `pos` was written
        assert not q[..., hd:].any() and not kw[..., hd:].any() and not vw[..., hd:].any(), s   # pad lanes stay zero
        return dict(q=q[..., :hd].contiguous(), k=kw[..., :hd].contiguous(), v=vw[..., :hd].contiguous(), swiglu=gp, logits=rows)

    def run_plain(s):
        out = {}
        for name, wp_, xin, K in (("wo", wop, ap, d), ("w2", w2p, gpin, F)):
            hp = hp0.clone()
            so = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
            rc = lib.lgen_gemm_tile(L.ptr(wp_), L.ptr(xin), L.ptr(hp), M, mts, d, K, L.EPI_RES, L.BF16, *s, 0, 0, 0, 0.0, L.ptr(so),
                                    L.stream())
            torch.cuda.synchronize()
            if rc == L.ERR_UNSUPPORTED:
                return None
            assert rc == 0, (s, rc)
            out[name], out[name + "_ssq"] = hp, so[:, : d // 16].clone()
        return out

    # the kernel's own normalised activations through an identity weight (exact): bit-identical to the skinny kernels' (same scales)
    eyep = pack_weight(torch.eye(d, dtype=dt).to(dev))
    xr_t = torch.zeros(mts * 16, d, dtype=dt, device=dev)
    xr_s = torch.zeros(mts * 16, d, dtype=dt, device=dev)
    L.check(lib.lgen_gemm_tile(L.ptr(eyep), L.ptr(xp), L.ptr(xr_t), M, mts, d, d, L.EPI_ROWS, L.BF16, 4, 1, 1, 4, 4, 4, 4, L.ptr(nw_d),
                               L.ptr(ssq), d // 16, 1e-5, 0, L.stream()), "tile norm + identity")
    L.check(lib.lgen_gemm(L.ptr(eyep), L.ptr(xp), L.ptr(xr_s), M, mts, d, d, L.EPI_ROWS, L.BF16, 2, 4, 8, L.ptr(nw_d), L.ptr(ssq),
                          d // 16, 1e-5, 0, 1, L.stream()), "skinny norm + identity")
    assert torch.equal(xr_t, xr_s)
    xn = xr_t[:M].float().cpu()
    xn_ref = O.rms_norm(x.float(), nw, 1e-5, dt)
    dx = (xn - xn_ref).abs()
    assert (dx > 0).float().mean().item() < 1e-3 and (dx <= torch.maximum(xn.abs(), xn_ref.abs()) * 2.0 ** -6 * 1.01).all()

    qkv = O.linear(xn, wq.float(), dt)
    xq, xk, xv = qkv.split([d, d, d], dim=-1)
    fr = freqs[pos:pos + 1]
    ref_q = O.apply_rotary_emb(xq.reshape(M, 1, H, hd), fr, dt)[:, 0]
    ref_k = O.apply_rotary_emb(xk.reshape(M, 1, H, hd), fr, dt)[:, 0]
    a1, a3 = O.linear(xn, w1.float(), dt), O.linear(xn, w3.float(), dt)
    ref_gp = O._rnd(O._rnd(torch.nn.functional.silu(a1), dt) * a3, dt)
    ref_rows = O.linear(xn, wh.float(), dt)
    first, nrun = None, 0
    for s in TILE_NORM:
        got = run_norm(s)
        if got is None:      # shape does not divide this M / exceeds the LDS budget: the library says so, nothing ran
            continue
        nrun += 1
        if first is None:
            first = got
            _close(got["q"][:M], ref_q, dt, f"q {s}", frac_ulp1=0.05, mag=qkv[:, :d].reshape(M, H, hd))
            _close(got["k"], ref_k, dt, f"k row {s}", frac_ulp1=0.05, mag=qkv[:, d:2 * d].reshape(M, H, hd))
            _close(got["v"], xv.reshape(M, H, hd), dt, f"v row {s}", frac_ulp1=0.05)
            _close(got["logits"][:M], ref_rows, dt, f"norm+lm_head {s}", frac_ulp1=0.05)
        for name in ("q", "k", "v", "logits", "swiglu"):
            if got[name] is None:
                continue
            if first[name] is None:
                first[name] = got[name]
                _close(unpack_act(got["swiglu"], M), ref_gp, dt, f"norm+swiglu {s}", frac_ulp1=0.08, ulps=3)
            assert torch.equal(first[name], got[name]), (s, name, (first[name].float() - got[name].float()).abs().max().item())
    assert nrun >= 4 and first["swiglu"] is not None, nrun
    # round 6: the statistics rows grouped ONCE into their four lane-group sums (lgen_ssq_group4; parts = 4 for the consumer) give the
    # SAME bits as the d / 16 partials -- every width, including GPT-3B's 200 partials, the case the engine groups
    ssq_g = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
    L.check(lib.lgen_ssq_group4(L.ptr(ssq), L.ptr(ssq_g), mts * 16, d // 16, L.stream()), "ssq_group4")
    assert lib.lgen_ssq_group4(L.ptr(ssq), L.ptr(ssq), mts * 16, d // 16, L.stream()) == L.ERR_BAD_ARG      # in place is refused
    assert lib.lgen_ssq_group4(L.ptr(ssq), L.ptr(ssq_g), mts * 16, 6, L.stream()) == L.ERR_BAD_ARG           # parts % 4
    ngrp = 0
    for s in TILE_NORM:
        got = run_norm(s, ssq_g, 4)
        if got is None:
            continue
        ngrp += 1
        for name in ("q", "k", "v", "logits", "swiglu"):
            if got[name] is not None:
                assert torch.equal(first[name], got[name]), ("grouped statistics", s, name)
    assert ngrp >= 4, ngrp

    firstp, nrun = None, 0
    for s in TILE_PLAIN:
        got = run_plain(s)
        if got is None:
            continue
        nrun += 1
        if firstp is None:
            firstp = got
            for name, w_, xin in (("wo", wo, a_in), ("w2", w2, g_in)):
                lin = O.linear(xin.float(), w_.float(), dt)
                ref = O._rnd(h0.float() + lin, dt)
                hrows = unpack_act(got[name], M)
                _close(hrows, ref, dt, f"{name}+res {s}", frac_ulp1=0.05, mag=lin)
                # statistics of the rows the kernel wrote: one fp32 partial per 16 columns, fixed order inside the tile
                part = hrows.float().cpu().pow(2).reshape(M, d // 16, 16).sum(-1)
                assert torch.allclose(got[name + "_ssq"][:M].cpu(), part, rtol=1e-5, atol=1e-6), name
        for name in firstp:
            assert torch.equal(firstp[name], got[name]), (s, name)
    assert nrun >= 3, nrun


@pytest.mark.parametrize("d,H,M,mt", [(1024, 16, 64, 1), (1280, 20, 64, 1), (1536, 24, 64, 1), (768, 12, 64, 1),
                                      (1024, 16, 128, 2), (1536, 24, 128, 2)])
@pytest.mark.parametrize("pos", [0, 301])
def test_qkv_fused_norm_rope_append_vs_oracle(d, H, M, mt, pos):
    """lgen_gemm_qkv_rope WITH the fused RMSNorm (norm_w / ssq_in): 64 rows with tiles (1, 4, 8) and 128 rows (two batches per
    chain, bench.py's default) with tiles (2, 4, 8); CPW 3 / 4 / 5 / 6 = GPT-B / L / XL / XXL, vs oracle rms_norm -> linear ->
    apply_rotary_emb -> KVCache.update."""
    from llamagen_amd.engine import pack_act, pack_weight, precompute_freqs_cis_2d
    from tests.test_gpu_gpt import _close, _rand
    L, dev = _L(), _dev()
    lib = L.lib()
    dt, hd, grid = torch.bfloat16, 64, 24
    S8 = O.find_multiple(1 + grid * grid, 8)
    mts = M // 16
    x = _rand((M, d), dt, 41, 1.3)
    w = _rand((3 * d, d), dt, 42, 0.03)
    nw = (1 + 0.1 * _rand((d,), torch.float32, 43)).to(dt)
    freqs = precompute_freqs_cis_2d(grid, hd, 10000.0, 1)
    xp, wp, nw_d, fr_d = pack_act(x.to(dev), mts), pack_weight(w.to(dev)), nw.to(dev), freqs.to(dev)
    ssq = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
    L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, L.BF16, L.stream()), "ssq_pack")
    kc = torch.zeros(M, H, S8, 64, dtype=dt, device=dev)
    vc = torch.zeros(M, H, S8, 64, dtype=dt, device=dev)
    q = torch.zeros(mts * 16, H, 64, dtype=dt, device=dev)
    state = torch.tensor([pos, 0], dtype=torch.int32, device=dev)
    L.check(lib.lgen_gemm_qkv_rope(L.ptr(wp), L.ptr(xp), L.ptr(q), L.ptr(kc), L.ptr(vc), L.ptr(fr_d), L.ptr(state), M, mts, d, H,
                                   hd, 64, S8, 0, L.BF16, mt, 4, 8, L.ptr(nw_d), L.ptr(ssq), d // 16, 1e-5, 1, L.stream()), "qkv fused")
    xn = O.rms_norm(x.float(), nw, 1e-5, dt)
    qkv = O.linear(xn, w.float(), dt)
    xq, xk, xv = qkv.split([d, d, d], dim=-1)
    fr = freqs[pos:pos + 1]
    xq = O.apply_rotary_emb(xq.reshape(M, 1, H, hd), fr, dt)[:, 0]
    xk = O.apply_rotary_emb(xk.reshape(M, 1, H, hd), fr, dt)[:, 0]
    _close(q[:M], xq, dt, "q", frac_ulp1=0.05, mag=qkv[:, :d].reshape(M, H, hd))
    _close(kc[:, :, pos], xk, dt, "k row", frac_ulp1=0.05, mag=qkv[:, d:2 * d].reshape(M, H, hd))
    _close(vc[:, :, pos], xv.reshape(M, H, hd), dt, "v row", frac_ulp1=0.05)
    kc[:, :, pos] = 0
    vc[:, :, pos] = 0
    assert not kc.any() and not vc.any()  # nothing but slot `pos` was written


def test_decode_code_batch32_384px_vs_oracle():
    """The decode_code() shape of the bench (32 x 24 x 24 codes -> 32 x 3 x 384 x 384): three of the images against
    the oracle decoder (GroupNorm statistics are per image, so the oracle decodes them one by one)."""
    from tests.cases import VQ_CASES
    m, sd = build_vq_holder(VQ_CASES["vq16_4x4"])
    dev = _dev()
    m = m.to(dev)
    codes = torch.randint(0, 16384, (32, 576), generator=torch.Generator().manual_seed(8))
    img = m.decode_code(codes.to(dev), [32, 8, 24, 24])
    assert tuple(img.shape) == (32, 3, 384, 384) and torch.isfinite(img).all()
    worst = 0.0
    for b in (0, 13, 31):
        ref = O.vq_decode_code(sd, codes[b:b + 1], [1, 8, 24, 24])
        worst = max(worst, (img[b:b + 1].cpu() - ref).abs().max().item())
    _log("decode_code_b32", dict(max_abs_err=worst))
    assert worst < 1e-3, worst  # north_star: decoded pixels within 1e-3 abs


def _pinned_model_schedule(e):
    """A chain of >= 256 rows of GPT-XXL / GPT-3B: RMSNorm fused, every decode GEMM on the tile family with the shapes
    engine.MODEL_TILE_SCHEDULES pins for that model (round 5: no on-device search, no timing-dependent family), and the engine
    reports the schedule as one an end-to-end test names -- this test."""
    from llamagen_amd.engine import MODEL_TILE_SCHEDULES, tile_schedule_key
    assert e.fuse_norm and e.tile_schedule_source() == "table"
    table = MODEL_TILE_SCHEDULES[(e.d, e.F, e.V)]
    t = table[tile_schedule_key(e.MTs, table)]
    sched = e.gemm_schedule()
    assert all(v["family"] == "tile" for v in sched.values()), sched
    want = {"wqkv": t["qkv"], "wo": t["wo"], "w13": t["w13"], "w2": t["w2"], "lm_head": t["head"]}
    assert want == {k: tuple(v["shape(wm,wn,mtv,ntv,kb,stages,lw)"]) for k, v in sched.items()}, sched
    assert e.tile_schedule_tested()
    return sched


@pytest.mark.parametrize("B", [64, 128])
def test_config4_gpt3b_shapes_bf16_vs_oracle(B):
    """BASELINE config 4 kernel shapes: GPT-3B widths (d 3200, 32 heads, head_dim 100 -> padded 128, F 8704),
    B = 64 -> 128 rows (MTs 8), 384 px (S8 584); depth cut to 4 layers so that the oracle finishes in seconds.
    B = 128: the 256-row chain `bench.py --config 4` runs (two batches of 64): RMSNorm fused (200 partial sums per row), the
    big-M tile family with the pinned GPT-3B shapes (engine.MODEL_TILE_SCHEDULES)."""
    kw = dict(n_layer=4, n_head=32, dim=3200, vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
              model_type="c2i")
    case = dict(kwargs=kw, wseed=22, lin_std=0.02)
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(4))
    recs, m = _teacher_forced(case, B, 4.0, early=3, late=[420], cond=cond)
    e = m._engine
    assert e.hd == 100 and e.hdp == 128 and e.MTs == B // 8 and e.F == 8704
    if B == 128:
        _log("config4_gpt3b_shapes_256rows", dict(schedule=str(_pinned_model_schedule(e))))
    _check(f"config4_gpt3b_shapes_b{B}", recs)


def test_config5_gptxl_t2i_shapes_bf16_vs_oracle():
    """BASELINE config 5 kernel shapes: GPT-XL widths (d 1280, 20 heads, F 3584), t2i with T = 120 caption tokens
    (caption_dim 2048, left-padded emb_masks folded into causal_mask through the drop-in API), 512 px
    (block_size 1024, S8 1144), B = 16 -> 32 rows, cfg 7.5; depth cut to 4 layers."""
    kw = dict(n_layer=4, n_head=20, dim=1280, vocab_size=16384, block_size=1024, cls_token_num=120, caption_dim=2048,
              model_type="t2i")
    case = dict(kwargs=kw, wseed=23, lin_std=0.02)
    B, T = 16, 120
    g = torch.Generator().manual_seed(5)
    emb = torch.randn(B, T, 2048, generator=g)
    lens = torch.randint(5, T + 1, (B,), generator=g)
    mask = torch.zeros(B, T, dtype=torch.int64)
    for b in range(B):
        mask[b, T - int(lens[b]):] = 1
    emb = (emb * mask[:, :, None]).to(torch.bfloat16).float()
    recs, m = _teacher_forced(case, B, 7.5, early=3, late=[700, 1142], cond=emb, emb_masks=mask, T=T)
    e = m._engine
    assert e.S8 == 1144 and e.T == 120 and e.MTs == 2
    _check("config5_gptxl_t2i_shapes", recs)


@pytest.mark.parametrize("B", [32, 128, 192])
def test_config3_gptxxl_shapes_bf16_vs_oracle(B):
    """BASELINE config 3 per-GPU kernel shapes: GPT-XXL widths (d 1536, 24 heads, F 4096; fused-norm CPW 6), B = 32 -> 64
    rows, 384 px; depth cut to 4 layers.  B = 128 / 192: chains of 256 / 384 rows (`bench.py --config 3` runs two chains of six
    batches = 384 rows) on the pinned GPT-XXL tile shapes."""
    kw = dict(n_layer=4, n_head=24, dim=1536, vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1,
              model_type="c2i")
    case = dict(kwargs=kw, wseed=24, lin_std=0.02)
    cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(6))
    recs, m = _teacher_forced(case, B, 4.0, early=3, late=[575], cond=cond)
    assert m._engine.fuse_norm
    if B >= 128:   # tile family, pinned shapes
        _log(f"config3_gptxxl_shapes_{2 * B}rows", dict(schedule=str(_pinned_model_schedule(m._engine))))
    _check(f"config3_gptxxl_shapes_b{B}", recs)


@pytest.mark.parametrize("model", ["XXL", "3B", "XL-t2i", "XL-t2i-384"])
def test_configs_3_4_5_wide_chain_shapes_bf16_vs_oracle(model):
    """Round 6: the chain widths `bench.py --config 3 / 4 / 5` runs now -- GPT-XXL two chains of eight batches of 32 (512 rows), GPT-3B
    two chains of four batches of 64 (512 rows), GPT-XL t2i two chains of eight batches of 16 (256 rows, T = 120 caption tokens with
    left-padded emb_masks) -- on the shapes engine.MODEL_TILE_SCHEDULES pins for that model and width (keys 32 / 32 / 16 and 24, measured with
    tools/gemm_tile_sweep.py).  Depth cut to 2 layers and the token grid to 16 x 16 (block_size 256; the GEMM shapes do not depend on
    either) so that two CPU oracles of 512 rows stay within ~15 GB and a minute; the 384 / 512 px slab lengths are held by the 128- /
    192- / 32-row cases above and the full-depth goldens below."""
    if model.startswith("XL-t2i"):
        kw = dict(n_layer=2, n_head=20, dim=1280, vocab_size=16384, block_size=256, cls_token_num=120, caption_dim=2048, model_type="t2i")
        # 256 rows (key 16), or the 384 rows bench.py --config 5 runs (2 x 12 batches of 16: key 24)
        B, T, scale, key, late = (192, 120, 7.5, 24, [370]) if model.endswith("384") else (128, 120, 7.5, 16, [370])
        g = torch.Generator().manual_seed(15)
        emb = torch.randn(B, T, 2048, generator=g)
        lens = torch.randint(5, T + 1, (B,), generator=g)
        mask = torch.zeros(B, T, dtype=torch.int64)
        for b in range(B):
            mask[b, T - int(lens[b]):] = 1
        cond = (emb * mask[:, :, None]).to(torch.bfloat16).float()
    else:
        H, d = {"XXL": (24, 1536), "3B": (32, 3200)}[model]
        kw = dict(n_layer=2, n_head=H, dim=d, vocab_size=16384, block_size=256, num_classes=1000, cls_token_num=1, model_type="c2i")
        B, T, scale, key, late, mask = 256, 1, 4.0, 32, [250], None
        cond = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(16))
    case = dict(kwargs=kw, wseed=26, lin_std=0.02)
    recs, m = _teacher_forced(case, B, scale, early=2, late=late, cond=cond, emb_masks=mask, T=T)
    e = m._engine
    from llamagen_amd.engine import MODEL_TILE_SCHEDULES, tile_schedule_key
    assert e.MTs == 2 * B // 16 and tile_schedule_key(e.MTs, MODEL_TILE_SCHEDULES[(e.d, e.F, e.V)]) == key
    _log(f"config_wide_chain_{model}_{2 * B}rows", dict(schedule=str(_pinned_model_schedule(e))))
    _check(f"config_wide_chain_{model}", recs)


@pytest.mark.parametrize("name", ["gptxxl_c3", "gpt3b_c4", "gptxl_t2i_c5"])
def test_configs_3_4_5_full_depth_vs_reference_golden(name):
    """BASELINE configs 3 / 4 / 5 at FULL depth (GPT-XXL 48 layers, GPT-3B 24 with head_dim 100, GPT-XL t2i 36 with T = 120 and
    emb_masks written through model.causal_mask), bf16, two images per batch: the HIP path's CFG-mixed logits of the prefill, two
    early positions and one late position on injected cache contents against the REFERENCE's (tests/golden/fulldepth_*.npz, made
    by tests/golden/make_golden.py in the build container) -- no CPU model runs here, so this is part of the default GPU suite
    (round 3 needed minutes of CPU oracle per model and was opt-in).  Bars: tests/cases.py (2 x the oracle-vs-reference distance)."""
    from tests.cases import FULL_DEPTH_CASES, cache_fill, full_depth_inputs
    from tests.test_oracle_golden import check_full_depth, full_depth_distances
    from tests.util import load_golden
    case = FULL_DEPTH_CASES[name]
    gold = load_golden("fulldepth_" + name)
    kw = case["kwargs"]
    dev, dt = _dev(), torch.bfloat16
    m, sd = build_gpt_holder(case)
    m = m.to(device=dev, dtype=dt)
    assert len(m.layers) == case["layers"]
    cond, emb_masks, steps = full_depth_inputs(case)
    B, T, N = case["batch"], kw["cls_token_num"], kw["block_size"]
    if kw["model_type"] == "c2i":
        cond_dev = torch.cat([cond, torch.ones_like(cond) * kw["num_classes"]]).to(dev)
    else:
        cond_dev = torch.cat([cond, torch.zeros_like(cond) + sd["cls_embedding.uncond_embedding"]]).to(dev).to(dt)
    m.setup_caches(2 * B, T + N, dt)
    if emb_masks is not None:   # generate.py:154-163 through the drop-in API
        em = torch.cat([emb_masks, emb_masks])
        cmd = m.causal_mask
        cmd[:, :, :T] = cmd[:, :, :T] & (em.to(dev).unsqueeze(1) != 0)
        cmd |= torch.eye(cmd.size(1), dtype=torch.bool, device=dev)
    e = m._engine
    got = {}
    for label, tok, ipos in steps:
        if label.startswith("late"):
            p = int(ipos[0])
            for li in range(e.L):
                kf, vf = cache_fill(2 * B, e.H, e.S8, e.hd, p, li, dt, seed=1000 + p)
                e.k_cache[li][:, :, :p, :e.hd] = kf.to(dev)
                e.v_cache[li][:, :, :p, :e.hd] = vf.to(dev)
        if tok is None:
            lg, _ = m(None, cond_dev, ipos.to(dev))
        else:
            lg, _ = m(torch.cat([tok, tok]).to(dev), None, ipos.to(dev).to(torch.int))
        got[label] = O.cfg_mix(lg[:, -1].float().cpu(), case["cfg_scale"]).numpy()
    dist = full_depth_distances(got, gold)
    for label, (emax, emean) in dist.items():
        _log("full_depth_" + name, dict(step=label, err_max=emax, err_mean=emean))
    check_full_depth(case, dist)
