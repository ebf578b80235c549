"""Host glue of llamagen_amd.generate (no GPU): CFG row doubling, the unconditional twins, cache sizing, caption-mask doubling
and argument checks -- what reference generate.py:126-163 does before the first forward -- with the engine replaced by a
stand-in that records what it is handed."""
import types

import pytest
import torch

from llamagen_amd.generate import generate


class _Engine:
    def __init__(self):
        self.call = None
        self.events = []

    def draw_noise(self, N, B, b0, n):
        self.events.append(("noise", N, B, b0, n))

    def generate_iter(self, model, rows, B, N, masks, sp):
        self.call = dict(rows=rows, B=B, N=N, masks=masks, sp=sp)
        yield
        return torch.zeros(B, N, dtype=torch.int32)


def _model(kind, T=1):
    m = types.SimpleNamespace(model_type=kind, num_classes=1000, _engine=_Engine(), caches=None,
                              tok_embeddings=types.SimpleNamespace(weight=torch.zeros(1, dtype=torch.bfloat16)),
                              cls_embedding=types.SimpleNamespace(uncond_embedding=torch.arange(T * 4, dtype=torch.float32).reshape(T, 4)))
    m.setup_caches = lambda max_batch_size, max_seq_length, dtype: setattr(m, "caches", (max_batch_size, max_seq_length, dtype))
    return m


def test_c2i_cfg_doubles_rows_with_the_null_class():
    m = _model("c2i")
    cond = torch.tensor([3, 999, 0])
    out = generate(m, cond, 16, cfg_scale=4.0, cfg_interval=5, temperature=0.9, top_k=7, top_p=0.8, sample_logits=False)
    c = m._engine.call
    assert tuple(out.shape) == (3, 16) and c["B"] == 3 and c["N"] == 16 and c["masks"] is None
    assert c["rows"].tolist() == [3, 999, 0, 1000, 1000, 1000] and c["rows"].dtype == cond.dtype
    assert m.caches == (6, 17, torch.bfloat16)
    assert c["sp"] == dict(use_cfg=True, cfg_scale=4.0, cfg_interval=5, temperature=0.9, top_k=7, top_p=0.8, sample_logits=False)


def test_c2i_without_cfg_passes_the_labels_through():
    m = _model("c2i")
    cond = torch.tensor([5, 6])
    generate(m, cond, 8, cfg_scale=1.0)
    c = m._engine.call
    assert c["rows"] is cond and m.caches == (2, 9, torch.bfloat16) and c["sp"]["use_cfg"] is False
    assert c["sp"]["top_k"] == 0 and c["sp"]["top_p"] == 1.0 and c["sp"]["temperature"] == 1.0 and c["sp"]["sample_logits"] is True


def test_t2i_cfg_twins_are_the_null_caption_and_masks_double():
    T = 3
    m = _model("t2i", T)
    cond = torch.randn(2, T, 4).to(torch.bfloat16)
    masks = torch.tensor([[0, 1, 1], [1, 1, 1]])
    generate(m, cond, 10, emb_masks=masks, cfg_scale=7.5)
    c = m._engine.call
    assert tuple(c["rows"].shape) == (4, T, 4) and c["rows"].dtype == torch.bfloat16
    assert torch.equal(c["rows"][:2], cond)
    null = m.cls_embedding.uncond_embedding.to(torch.bfloat16)
    assert torch.equal(c["rows"][2], null) and torch.equal(c["rows"][3], null)
    assert torch.equal(c["masks"], torch.cat([masks, masks])) and m.caches == (4, T + 10, torch.bfloat16)


def test_argument_errors_follow_the_reference():
    with pytest.raises(Exception, match="please check model type"):
        generate(_model("i2i"), torch.tensor([1]), 4)
    m = _model("t2i", 3)
    with pytest.raises(AssertionError):
        generate(m, torch.zeros(2, 3, 4), 4, emb_masks=torch.ones(1, 3))
    with pytest.raises(AssertionError):
        generate(m, torch.zeros(2, 3, 4), 4, emb_masks=torch.ones(2, 5))
    with pytest.raises(TypeError):
        generate(_model("c2i"), torch.tensor([1]), 4, topk=3)


def test_chain_of_batches_draws_labels_and_noise_in_consecutive_generate_order():
    """pipeline hook `_more_conds` (several batches on one decode chain): batch j's labels are evaluated, then batch j's noise
    is drawn, THEN batch j + 1's labels -- the order consecutive reference generate() calls consume the default generator
    (sample_c2i_ddp.py:128-140) -- and the chain's rows are [all conditional rows, then all their null twins]."""
    m = _model("c2i")
    ev = m._engine.events

    def batch(i):
        def f():
            ev.append(("labels", i))
            return torch.tensor([10 * i + 1, 10 * i + 2])
        return f
    first = batch(0)()
    generate(m, first, 6, cfg_scale=4.0, _more_conds=[batch(1), batch(2)])
    assert ev == [("labels", 0), ("noise", 6, 6, 0, 2), ("labels", 1), ("noise", 6, 6, 2, 2), ("labels", 2), ("noise", 6, 6, 4, 2)]
    c = m._engine.call
    assert c["B"] == 6 and c["rows"].tolist() == [1, 2, 11, 12, 21, 22] + [1000] * 6 and m.caches == (12, 7, torch.bfloat16)
    assert c["sp"]["_noise_prefilled"] is True
    ev.clear()
    generate(m, torch.tensor([1, 2]), 6, cfg_scale=1.0, sample_logits=False, _more_conds=[torch.tensor([3, 4])])   # greedy: no draws
    assert ev == [] and m._engine.call["rows"].tolist() == [1, 2, 3, 4] and "_noise_prefilled" not in m._engine.call["sp"]
    with pytest.raises(ValueError):
        generate(m, torch.tensor([1, 2]), 6, _more_conds=[torch.tensor([3, 4, 5])])
    # text-conditional chains: captions and their masks are concatenated batch by batch (a batch without its own mask takes the shared one)
    mt = _model("t2i", 3)
    c0, c1 = torch.ones(2, 3, 4), 2 * torch.ones(2, 3, 4)
    m0, m1 = torch.tensor([[0, 1, 1], [1, 1, 1]]), torch.tensor([[0, 0, 1], [0, 1, 1]])
    generate(mt, c0, 4, emb_masks=m0, cfg_scale=7.5, _more_conds=[lambda: (c1, m1)])
    c = mt._engine.call
    assert tuple(c["rows"].shape) == (8, 3, 4) and torch.equal(c["rows"][:4], torch.cat([c0, c1]))
    assert torch.equal(c["masks"], torch.cat([m0, m1, m0, m1])) and mt.caches == (8, 3 + 4, torch.bfloat16)
    generate(mt, c0, 4, emb_masks=m0, cfg_scale=1.0, _more_conds=[c1])
    assert torch.equal(mt._engine.call["masks"], torch.cat([m0, m0]))
    with pytest.raises(ValueError):
        generate(mt, c0, 4, cfg_scale=1.0, _more_conds=[(c1, m1)])
