"""Host logic of llamagen_amd.pipeline.SamplingPipeline.run (no GPU): which batches share a decode chain, the padding of an
incomplete last chain, the per-batch hand-back (on_done once per batch, in the reference's batch order) and lane refill.
The lanes are stand-ins that "generate" ids = label * 1000 + token position, so every output row names the input row it
came from."""
import torch

from llamagen_amd.generate import PadBatch
from llamagen_amd.pipeline import SamplingPipeline


class _Lane:
    def __init__(self, steps):
        self.stream = self.vq_stream = None
        self.steps, self.left, self.job, self.started = steps, 0, None, []

    @property
    def busy(self):
        return self.job is not None

    def start(self, job_id, cond, max_new_tokens, decode_shape, gen_kw):
        # what generate_iter does with `_more_conds` (the other batches of the chain, evaluated in RNG order after the first)
        # (PadBatch: filler of an incomplete last chain -- valid conditioning, no noise draws, rows dropped)
        more = gen_kw.get("_more_conds") or []
        self.pads = sum(isinstance(c, PadBatch) for c in more)
        parts = [cond] + [c.value[0] if isinstance(c, PadBatch) else (c() if callable(c) else c) for c in more]
        if any(c.shape != cond.shape for c in parts):
            raise ValueError("batches that share a chain must have the same size")
        cond = torch.cat(parts)
        self.job, self.left = (job_id, cond.clone(), max_new_tokens, decode_shape), self.steps
        self.started.append((job_id, cond.clone(), decode_shape))

    def advance(self, nsteps=1):
        self.left -= nsteps
        if self.left > 0:
            return None
        job_id, cond, N, shape = self.job
        self.job = None
        ids = cond[:, None] * 1000 + torch.arange(N)[None]
        img = ids[:, :1].float().reshape(-1, 1, 1, 1).expand(-1, 3, 2, 2)
        return job_id, ids, img


class _Pipe(SamplingPipeline):
    def __init__(self, lanes, bpc):  # the real constructor creates HIP streams and decode engines
        self.dev, self.lanes, self.steps_per_turn, self.vq_stream, self.bpc = torch.device("cpu"), lanes, 1, None, bpc

    def _join(self):
        pass


def _conds(k, rows=3):
    return [torch.arange(rows) + 10 * i for i in range(k)]


def test_one_batch_per_chain_keeps_submission_order():
    lanes = [_Lane(4), _Lane(2)]
    out = _Pipe(lanes, 1).run(_conds(5), 6)
    assert [int(ids[0, 0]) // 1000 for ids, _ in out] == [0, 10, 20, 30, 40]
    assert [j[0][0] for lane in lanes for j in lane.started] != []  # both lanes took work
    assert sorted(j[0][0] for lane in lanes for j in lane.started) == [0, 1, 2, 3, 4]
    for ids, img in out:
        assert tuple(ids.shape) == (3, 6) and tuple(img.shape) == (3, 3, 2, 2)


def test_two_batches_per_chain_group_pad_and_split():
    lanes = [_Lane(3), _Lane(3)]
    seen = []
    conds = _conds(5)
    out = _Pipe(lanes, 2).run(conds, 4, decode_shape=[3, 8, 2, 2], on_done=lambda j, ids, img: (seen.append(j), ids, img)[1:])
    chains = sorted((j for lane in lanes for j in lane.started), key=lambda t: t[0][0])
    assert [(c[0][0], c[0][1]) for c in chains] == [(0, 2), (2, 2), (4, 1)]  # (first batch, batches in the chain)
    for (first, n, rows), cond, shape in chains:
        assert rows == 3 and cond.shape[0] == 6 and shape == [6, 8, 2, 2]  # every chain has the full chain shape
        want = torch.cat([conds[first + i] for i in range(n)] + [conds[first]] * (2 - n))
        assert torch.equal(cond, want)  # an incomplete chain is filled up with PadBatch copies of its first batch (no noise drawn for them)
    assert sorted(seen) == [0, 1, 2, 3, 4] and len(out) == 5  # on_done once per BATCH; the padding is dropped
    for i, (ids, img) in enumerate(out):
        assert torch.equal(ids[:, 0] // 1000, conds[i]) and tuple(ids.shape) == (3, 4) and tuple(img.shape) == (3, 3, 2, 2)


def test_callable_conds_are_drawn_in_batch_order_and_sizes_must_match():
    order = []

    def mk(i):
        return lambda: (order.append(i), torch.arange(2) + 10 * i)[1]
    out = _Pipe([_Lane(1), _Lane(1)], 2).run([mk(i) for i in range(4)], 3)
    assert order == [0, 1, 2, 3] and [int(ids[0, 0]) // 1000 for ids, _ in out] == [0, 10, 20, 30]
    try:
        _Pipe([_Lane(1)], 2).run([torch.arange(2), torch.arange(3)], 3)
    except ValueError:
        pass
    else:
        raise AssertionError("batches of different sizes must not share a chain")


def test_prepare_gives_every_lane_one_whole_chain_c2i_and_t2i(monkeypatch):
    """prepare() = one throw-away chain PER LANE (slabs allocated, graph captured before any timed run), for class- and for
    text-conditional models alike: with batches_per_chain > 1 that is bpc conditioning batches per lane (round-3 advisor finding:
    the t2i branch used to submit one batch per lane, which run() packed into a single chain on lane 0)."""
    import types
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    for model_type in ("c2i", "t2i"):
        lanes = [_Lane(steps=2) for _ in range(3)]
        lanes[0].gpt = types.SimpleNamespace(model_type=model_type, num_classes=1000, cls_token_num=5,
                                             config=types.SimpleNamespace(caption_dim=8),
                                             tok_embeddings=types.SimpleNamespace(weight=torch.zeros(1, dtype=torch.float32)))
        pipe = _Pipe(lanes, bpc=4)
        seen = []

        def start(self, job_id, cond, max_new_tokens, decode_shape, gen_kw, _seen=seen):
            more = gen_kw.get("_more_conds") or []
            assert not any(isinstance(c, PadBatch) for c in more)          # whole chains: nothing to pad
            _seen.append((id(self), 1 + len(more), cond.shape[0]))
            self.job, self.left = (job_id, torch.zeros(cond.shape[0] * (1 + len(more)), dtype=torch.long), max_new_tokens, decode_shape), 1
        monkeypatch.setattr(_Lane, "start", start)
        pipe.prepare(batch=2, max_new_tokens=3)
        monkeypatch.undo()
        monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
        assert sorted(s[0] for s in seen) == sorted(id(l) for l in lanes), (model_type, seen)   # every lane ran exactly one chain
        assert all(nb == 4 and rows == 2 for _, nb, rows in seen), (model_type, seen)
