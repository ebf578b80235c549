"""Host logic of llamagen_amd.serve.ContinuousBatcher (no GPU): the request queue, slot refill between steps and the
"a request that entered at step k is complete after step k + N - 1" accounting.  The device step is a stand-in that writes
token = label * 1000 + step into the slot's sequence, exactly where lgen_sample_rows would."""
import collections
import types

import pytest
import torch

from llamagen_amd.serve import ContinuousBatcher, bucket_ladder, pick_bucket, plan_moves


class _Eng:  # the tensors _switch() moves between buckets
    def __init__(self, rows, S8):
        self.k_cache = torch.zeros(1, rows, 1, S8, 1)
        self.v_cache = torch.zeros(1, rows, 1, S8, 1)
        self.cur_tok = torch.zeros(rows, dtype=torch.int32)


class _Batcher(ContinuousBatcher):
    def __init__(self, slots, N, cfg=True, buckets=None, shrink_after=2):  # the real constructor builds DecodeEngines on the GPU
        self.slots, self.N, self.use_cfg = slots, N, cfg
        self.num_classes, self.V, self.t2i, self.T = 1000, 32, False, 1
        self.dev = torch.device("cpu")
        self.bucket_sizes = bucket_ladder(slots, buckets)
        self.shrink_after, self._shrink_wait, self.switches, self._buckets = shrink_after, 0, 0, {}
        self.cur = self._bucket(self.bucket_sizes[0] if len(self.bucket_sizes) > 1 else slots)
        self.row_step = torch.full((slots,), N, dtype=torch.int32)
        self.noise = None
        self.seq = torch.zeros(slots, N, dtype=torch.int32)
        self._queue = collections.deque()
        self._slot_req, self._slot_left = [None] * slots, [0] * slots
        self._next_id, self.steps_run = 0, 0
        self.active_per_step, self.bucket_per_step = [], []

    def _bucket(self, size):
        if size not in self._buckets:
            rows = 2 * size if self.use_cfg else size
            self._buckets[size] = types.SimpleNamespace(B=size, B2=rows, eng=_Eng(rows, self.N + 1), graph=None,
                                                        row_pos=torch.full((rows,), 1, dtype=torch.int32),
                                                        cond=torch.full((rows,), 1000, dtype=torch.int32))
        return self._buckets[size]

    def _step(self):
        live, e = 0, self.eng
        for b in range(self.B):
            st = int(self.row_step[b])
            if st < self.N:  # an occupied slot: what the layer chain (K/V append) and the sampler kernel do for it
                label = int(self.cond[b])
                rows = [b] + ([self.B + b] if self.use_cfg else [])
                for r in rows:
                    assert int(self.row_pos[r]) == st
                    # the request's earlier K/V entries are in ITS rows of the current bucket, whatever bucket wrote them
                    assert e.k_cache[0, r, 0, :st, 0].tolist() == [label * 1000 + p + (0.5 if r != b else 0) for p in range(st)], (b, st)
                    e.k_cache[0, r, 0, st, 0] = label * 1000 + st + (0.5 if r != b else 0)
                    assert st == 0 or int(e.cur_tok[r]) == label * 1000 + st - 1
                    e.cur_tok[r] = label * 1000 + st
                    self.row_pos[r] += 1
                self.seq[b, st] = label * 1000 + st
                self.row_step[b] += 1
                live += 1
        self.active_per_step.append(live)
        self.bucket_per_step.append(self.B)


def test_requests_outnumbering_slots_are_all_served_in_full():
    cb = _Batcher(slots=3, N=5)
    labels = [7, 1, 9, 4, 2, 8, 3]
    rids = [cb.submit(l) for l in labels]
    done = cb.run(use_graph=False)
    assert sorted(done) == rids
    for rid, l in zip(rids, labels):
        assert done[rid].tolist() == [l * 1000 + s for s in range(5)]
    # 7 requests x 5 steps on 3 slots: slots are refilled the step after they finish, none idles while the queue has work
    assert cb.steps_run == len(cb.active_per_step) == 15 and sum(cb.active_per_step) == 35
    assert cb.active_per_step[:10] == [3] * 10
    assert all(r is None for r in cb._slot_req) and not cb._queue


def test_late_submissions_join_a_running_batch_without_cfg_rows():
    cb = _Batcher(slots=2, N=4, cfg=False)
    a = cb.submit(5)
    first = cb.run(use_graph=False)
    b, c, d = cb.submit(6), cb.submit(7), cb.submit(8)
    second = cb.run(use_graph=False)
    assert first[a].tolist() == [5000, 5001, 5002, 5003]
    assert sorted(second) == [b, c, d] and second[d].tolist() == [8000, 8001, 8002, 8003]
    assert cb.steps_run == 4 + 8


def test_submit_validates_label_and_noise():
    cb = _Batcher(slots=2, N=4)
    with pytest.raises(IndexError):
        cb.submit(1001)            # the reference's embedding lookup raises for labels outside the table (gpt.py:78-83)
    with pytest.raises(IndexError):
        cb.submit(-1)
    cb.submit(1000)                # the null class is a valid table row
    with pytest.raises(ValueError):
        cb.submit(3, noise=torch.ones(4, 32))   # greedy batcher (no noise buffer)
    cb.noise = torch.zeros(2, 4, 32)
    with pytest.raises(ValueError):
        cb.submit(3, noise=torch.ones(5, 32))
    cb.submit(3, noise=torch.ones(4, 32))


def test_bucket_ladder_choice_and_move_plan():
    assert bucket_ladder(64, None) == [64] and bucket_ladder(64, (32, 16, 16, 64)) == [16, 32, 64]
    with pytest.raises(ValueError):
        bucket_ladder(64, (0, 16))
    with pytest.raises(ValueError):
        bucket_ladder(64, (128,))
    assert [pick_bucket([16, 32, 64], n) for n in (0, 1, 16, 17, 32, 33, 64, 500)] == [16, 16, 16, 32, 32, 64, 64, 64]
    # growing keeps every index; shrinking keeps the low ones and fills the lowest free slots with the rest
    assert plan_moves([True, False, True, True], 8) == [(0, 0), (2, 2), (3, 3)]
    assert plan_moves([False, True, False, False, True, False, True, False], 4) == [(1, 1), (4, 0), (6, 2)]
    with pytest.raises(ValueError):
        plan_moves([True, True, True], 2)


@pytest.mark.parametrize("cfg", [True, False])
def test_buckets_grow_and_shrink_with_load_and_requests_keep_their_state(cfg):
    """slots=8 with captured slot counts 2 / 4 / 8: one request alone runs in the 2-slot bucket, a burst grows the batch at once,
    the tail shrinks it again after `shrink_after` steps; every request's K/V history, token and counters follow it through every
    switch (the stand-in step asserts that before each token), so every sequence comes out complete."""
    cb = _Batcher(slots=8, N=6, cfg=cfg, buckets=(2, 4), shrink_after=2)
    first = cb.submit(5)
    out = {}
    cb._load(0, *cb._queue.popleft())
    for _ in range(2):           # two steps alone in the smallest bucket
        cb._step(); cb.steps_run += 1; cb._account(out)
    assert cb.bucket_per_step == [2, 2]
    labels = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20]
    rids = [cb.submit(l) for l in labels[:6]]
    out.update(cb.run(use_graph=False))          # 7 live -> the 8-slot bucket at once
    rids += [cb.submit(l) for l in labels[6:9]]
    out.update(cb.run(use_graph=False))          # 3 live -> 4 slots after shrink_after steps (requests move mid-sequence)
    rids.append(cb.submit(labels[9]))
    out.update(cb.run(use_graph=False))          # 1 live -> 2 slots
    assert sorted(out) == [first] + rids
    for rid, l in zip([first] + rids, [5] + labels):
        assert out[rid].tolist() == [l * 1000 + s for s in range(6)]
    assert cb.bucket_per_step[2] == 8 and 4 in cb.bucket_per_step and cb.bucket_per_step[-1] == 2
    assert cb.switches >= 3 and all(r is None for r in cb._slot_req)
    # no step ran more slots than its bucket holds, and every switch down waited for shrink_after steps of lower load
    assert all(a <= b for a, b in zip(cb.active_per_step, cb.bucket_per_step))


def test_single_bucket_never_switches():
    cb = _Batcher(slots=3, N=4)
    for l in (1, 2, 3, 4):
        cb.submit(l)
    cb.run(use_graph=False)
    assert cb.switches == 0 and set(cb.bucket_per_step) == {3}


@pytest.mark.parametrize("cfg", [True, False])
def test_shrinking_relocates_requests_from_high_slots(cfg):
    """Two requests running in slots 5 and 6 of the 8-slot bucket: after `shrink_after` steps they continue in slots 0 and 1 of
    the 2-slot bucket with their K/V rows, tokens, positions, class ids and token rows moved (the step asserts the history)."""
    cb = _Batcher(slots=8, N=7, cfg=cfg, buckets=(2,), shrink_after=3)
    cb._switch(cb._bucket(8))
    a, b = cb.submit(41), cb.submit(42)
    cb._load(5, *cb._queue.popleft())
    cb._load(6, *cb._queue.popleft())
    out = cb.run(use_graph=False)
    assert out[a].tolist() == [41000 + s for s in range(7)] and out[b].tolist() == [42000 + s for s in range(7)]
    assert cb.bucket_per_step == [8, 8, 2, 2, 2, 2, 2] and cb.switches == 2
    assert cb.row_step.tolist()[2:] == [7] * 6        # the vacated slots read as empty
