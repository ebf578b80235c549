"""Host logic of llamagen_amd.serve.ContinuousBatcher (no GPU): the request queue, slot refill between steps and the
"a request that entered at step k is complete after step k + N - 1" accounting.  The device step is a stand-in that writes
token = label * 1000 + step into the slot's sequence, exactly where lgen_sample_rows would."""
import collections

import torch

from llamagen_amd.serve import ContinuousBatcher


class _Batcher(ContinuousBatcher):
    def __init__(self, slots, N, cfg=True):  # the real constructor builds a DecodeEngine on the GPU
        self.B, self.N, self.use_cfg = slots, N, cfg
        self.num_classes, self.V, self.t2i, self.T = 1000, 32, False, 1
        self.B2 = 2 * slots if cfg else slots
        R = 16
        self.dev = torch.device("cpu")
        self.row_pos = torch.full((R,), N, dtype=torch.int32)
        self.row_step = torch.full((slots,), N, dtype=torch.int32)
        self.cond = torch.full((R,), 1000, dtype=torch.int32)
        self.noise = None
        self.seq = torch.zeros(slots, N, dtype=torch.int32)
        self._graph, self._queue = None, collections.deque()
        self._slot_req, self._slot_left = [None] * slots, [0] * slots
        self._next_id, self.steps_run = 0, 0
        self.active_per_step = []

    def _step(self):
        live = 0
        for b in range(self.B):
            st = int(self.row_step[b])
            if st < self.N:  # an occupied slot: what the sampler kernel does for it
                assert int(self.row_pos[b]) == st and (not self.use_cfg or int(self.row_pos[self.B + b]) == st)
                self.seq[b, st] = int(self.cond[b]) * 1000 + st
                self.row_step[b] += 1
                self.row_pos[b] += 1
                if self.use_cfg:
                    self.row_pos[self.B + b] += 1
                live += 1
        self.active_per_step.append(live)


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
    import pytest
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
