"""Batch-level pipelining of the sampling hot path on one GPU.

Every kernel of the KV-cached decode chain is latency-bound at the reference's batch sizes (a 64-row
GEMM over a 6 MB weight matrix takes ~5 us of which < 1 us is HBM time), so one batch leaves most of
the chip idle most of the time.  Independent batches are independent units (own labels, own KV cache,
own RNG draws -- the same observation sample_c2i_ddp.py:114-157 uses across GPUs), so this module keeps
`lanes` batches in flight on separate HIP streams: each lane owns a decode engine (KV slabs,
workspaces, captured decode-step hipGraph) and shares the packed weights.  The host never waits for
the GPU; it submits the lanes' decode steps ROUND-ROBIN (a lane's hardware queue only holds a few
dozen step graphs, so submitting one batch's 576 steps in one go would stall the host and serialise
the lanes), and the lanes' kernels interleave on the device (measured on MI355X, GPT-L, 32 images per
batch: 2 lanes = 1.5x the decode throughput of 1).

Reference call sites this stands in for: the per-iteration body of sample_c2i_ddp.py:128-143 /
sample_c2i.py:82-92 (generate -> decode_code), issued for consecutive iterations.  Noise is drawn from
the default generator batch by batch in submission order, exactly as consecutive generate() calls would.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .generate import PadBatch, generate_iter


def _eval_batch(c):
    """One entry of SamplingPipeline.run(conds): a conditioning tensor, a (caption_embs, emb_masks) tuple, or a callable that
    returns either (evaluated only when its chain starts: RNG order) -> (cond, emb_masks or None)."""
    c = c() if callable(c) else c
    return c if isinstance(c, tuple) else (c, None)


class SamplingLane:
    """One in-flight batch: generate() and decode_code() on the lane's stream.  `vq_stream` (optional) moves the
    VQ decoder to a separate (shared, lower-priority) stream so that the lane can start its next batch at once --
    measured WORSE on MI355X (50 vs 65 img/s with 3 lanes: a 4th active stream drops the dispatch rate, see
    tools/ubench_cp.py), so the default keeps the decoder on the lane's stream."""

    def __init__(self, gpt, vq=None, stream: Optional[torch.cuda.Stream] = None, primary: bool = False,
                 vq_stream: Optional[torch.cuda.Stream] = None, vq_chunk: int = 0):
        self.gpt = gpt if primary else gpt.lane_view()
        self.vq = vq
        self.dev = next(gpt.parameters()).device
        self.stream = stream or torch.cuda.Stream(device=self.dev)
        self.vq_stream = vq_stream or self.stream
        self.vq_chunk = int(vq_chunk)   # > 0: decode_code() in pieces of this many images (a chain of several batches decodes
        self._it = None                 # batch by batch: activation tensors stay at the per-batch size, 2.4 GB at 32 x 384 px)
        self._job = None

    @property
    def busy(self) -> bool:
        return self._it is not None

    def start(self, job_id, cond, max_new_tokens, decode_shape, gen_kw):
        cur = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(cur)  # inputs produced on the caller's stream
        self._job = (job_id, cond.shape[0] * (1 + len(gen_kw.get("_more_conds") or ())), max_new_tokens, decode_shape)
        with torch.cuda.stream(self.stream), torch.no_grad():
            self._it = generate_iter(self.gpt, cond, max_new_tokens, **gen_kw)
            next(self._it)  # noise draws + prefill + first token

    def advance(self, nsteps: int = 1):
        """Enqueue up to nsteps more decode steps; returns None while the batch is unfinished, else
        (job_id, ids, images) with decode_code() enqueued behind the last step."""
        with torch.cuda.stream(self.stream), torch.no_grad():
            try:
                for _ in range(nsteps):
                    next(self._it)
                return None
            except StopIteration as stop:
                idx = stop.value
        job_id, B, N, shape = self._job
        img = None
        if self.vq is not None:
            self.vq_stream.wait_stream(self.stream)
            idx.record_stream(self.vq_stream)
            with torch.cuda.stream(self.vq_stream), torch.no_grad():
                lat = int(round(N ** 0.5))
                shp = list(shape or [B, 8, lat, lat])
                if self.vq_chunk > 0 and B > self.vq_chunk:
                    img = torch.cat([self.vq.decode_code(idx[i:i + self.vq_chunk], [min(self.vq_chunk, B - i)] + shp[1:])
                                     for i in range(0, B, self.vq_chunk)])
                else:
                    img = self.vq.decode_code(idx, shp)
        self._it = None
        cur = torch.cuda.current_stream(self.dev)
        idx.record_stream(cur)
        if img is not None:
            img.record_stream(cur)
        return job_id, idx, img


class SamplingPipeline:
    """Keeps up to `lanes` batches in flight.  run(conds) -> [(ids, images)] in submission order; the
    results are enqueued-behind on the CURRENT stream when run() returns (no host synchronisation)."""

    def __init__(self, gpt, vq=None, lanes: int = 2, steps_per_turn: int = 1, vq_low_priority: bool = False,
                 batches_per_chain: int = 1, vq_chunk: int = 0):
        self.dev = next(gpt.parameters()).device
        # `batches_per_chain` consecutive batches of run() share ONE decode chain (their rows are concatenated): rows never
        # interact before the sampler and the sampler pairs row b with row B + b only, so every image is what its own
        # generate() call would have produced from the same noise, while the chain's fixed costs (a dependent launch per
        # kernel, the weight stream of every GEMM) are paid once for twice the rows.  Measured on MI355X, GPT-L 384 px,
        # 32 images per batch: one chain in flight 47 -> 61-70 img/s, 3 chains in flight 76 -> 85+ (tools/exp_r2e.py).
        self.bpc = max(1, int(batches_per_chain))
        # (text-conditional batches may be given as (caption_embs, emb_masks) tuples: a chain concatenates both)
        # optional: one shared stream for every lane's VQ decode (see SamplingLane)
        self.vq_stream = torch.cuda.Stream(device=self.dev, priority=0) if (vq is not None and vq_low_priority) else None
        lanes = max(1, lanes)
        # (CU-masked lane streams and a CU-confined decoder stream were measured slower in rounds 1-3 -- 69 vs 91 img/s, 45-59 vs 76 --
        # and removed in round 4: profiles/HISTORY.md)
        self.lanes: List[SamplingLane] = [SamplingLane(gpt, vq, stream=None, primary=(i == 0), vq_stream=self.vq_stream,
                                                       vq_chunk=vq_chunk) for i in range(lanes)]
        self.steps_per_turn = steps_per_turn

    def prepare(self, batch: int, max_new_tokens: int, **gen_kw):
        """One throw-away batch per lane: allocates the lane's KV slabs / workspaces and captures its decode
        graph, so that later runs only enqueue."""
        g = torch.Generator(device="cpu").manual_seed(0)
        gpt = self.lanes[0].gpt
        conds = []
        for _ in self.lanes:
            if gpt.model_type == "c2i":
                conds += [torch.randint(0, max(1, gpt.num_classes), (batch,), generator=g).to(self.dev) for _ in range(self.bpc)]
            else:   # one whole chain per lane here too: every lane allocates its slabs and captures its graph now, not in a timed run
                conds += [torch.zeros(batch, gpt.cls_token_num, gpt.config.caption_dim, device=self.dev,
                                      dtype=gpt.tok_embeddings.weight.dtype) for _ in range(self.bpc)]
        self.run(conds, max_new_tokens, **gen_kw)
        torch.cuda.synchronize(self.dev)

    def run(self, conds: Sequence[torch.Tensor], max_new_tokens: int, decode_shape=None, on_done=None,
            **gen_kw) -> List[Tuple[torch.Tensor, Optional[torch.Tensor]]]:
        """on_done(job_id, ids, images) (optional) is called as soon as a batch's decode_code() has been ENQUEUED, with
        that lane's stream current -- the place to enqueue per-batch post-processing and the step's collective
        (dist.gather_to_root) so that it overlaps the other lanes' decode steps; its return value replaces
        (ids, images) in the result list."""
        results = [None] * len(conds)
        nxt = 0
        while True:
            for lane in self.lanes:  # batches start in submission order (-> RNG consumption order)
                if not lane.busy and nxt < len(conds):
                    # a callable is evaluated only now, so that a driver can draw its labels from the device generator in the
                    # reference's order (labels of batch i, noise of batch i, labels of batch i+1, ...: sample_c2i_ddp.py:128-140)
                    group = list(conds[nxt:nxt + self.bpc])
                    ev = _eval_batch
                    chain, mask0 = ev(group[0])
                    rows = chain.shape[0]
                    kw = gen_kw if mask0 is None else dict(gen_kw, emb_masks=mask0)
                    if self.bpc == 1:
                        shape = decode_shape
                    else:
                        # the chain's first batch now; the others are evaluated by generate_iter in RNG order (labels of batch j,
                        # noise of batch j, labels of batch j + 1, ...).  A last, incomplete group repeats its last batch (rows
                        # computed and dropped): one chain shape per lane
                        # (PadBatch: rows computed and dropped, NO Exp(1) draws -- the default generator is left exactly where
                        # the reference's consecutive generate() calls would leave it)
                        pad = PadBatch((chain, mask0))
                        more = group[1:] + [pad] * (self.bpc - len(group))
                        if "_noise_seq" in gen_kw:  # injected noise (tests): evaluate now, the whole chain's block is given
                            parts = [(chain, mask0)] + [c.value if isinstance(c, PadBatch) else ev(c) for c in more]
                            if any(c.shape[0] != rows for c, _ in parts):
                                raise ValueError("batches that share a chain must have the same size")
                            chain = torch.cat([c for c, _ in parts])
                            shared = gen_kw.get("emb_masks")
                            if mask0 is not None or shared is not None:
                                kw = dict(gen_kw, emb_masks=torch.cat([m if m is not None else shared for _, m in parts]))
                        else:
                            kw = dict(kw, _more_conds=more)
                        shape = None if decode_shape is None else [rows * self.bpc] + list(decode_shape[1:])
                    lane.start((nxt, len(group), rows), chain, max_new_tokens, shape, kw)
                    nxt += len(group)
            active = [lane for lane in self.lanes if lane.busy]
            if not active:
                break
            for lane in active:
                done = lane.advance(self.steps_per_turn)
                if done is not None:
                    (first, n, rows), ids, img = done
                    for i in range(n):  # hand the chain's batches back one by one, in submission order
                        sl = slice(i * rows, (i + 1) * rows)
                        part = (first + i, ids[sl], None if img is None else img[sl])
                        if on_done is not None:
                            with torch.cuda.stream(lane.vq_stream), torch.no_grad():
                                results[first + i] = on_done(*part)
                        else:
                            results[first + i] = (part[1], part[2])
        self._join()
        return results

    def _join(self):
        """The caller's stream waits for every lane: results are enqueued-behind when run() returns."""
        cur = torch.cuda.current_stream(self.dev)
        for lane in self.lanes:
            cur.wait_stream(lane.stream)
        if self.vq_stream is not None:
            cur.wait_stream(self.vq_stream)
