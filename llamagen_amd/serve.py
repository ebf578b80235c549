"""Token-level continuous batching of class-conditional sampling on the HIP engine.

What it stands in for: the reference's vLLM fork (autoregressive/serve/llm.py:22-267 request queue + engine loop,
serve/sampler.py:54-58,106-108 paired conditional / unconditional sequences with CFG applied at sampling time,
serve/model_runner.py:36-41,982-1076 graph replay over a fixed batch).  Same idea, MI355X-native form:

  * a fixed number of SLOTS; slot b owns rows b (conditional) and B + b (unconditional) of the step batch, its own KV-cache
    rows and its own Exp(1) noise block -- nothing of one request touches another's rows;
  * ONE captured hipGraph per step batch: embed (per-row source) -> L x [norm+wqkv+RoPE+append | attention | wo | norm+w1,w3 |
    w2] -> norm+lm_head -> sample, all with PER-ROW positions (`row_pos`, device ints): a row at position 0 is a fresh
    request and takes its class embedding, any other row continues with the token it sampled in the previous step;
  * the host refills finished slots between replays (a handful of scalar writes per new request, no device sync: a request
    that entered at replay k is complete after replay k + N - 1, which the host knows without asking the GPU).

Text-conditional requests (round 3) bring a [T, caption_dim] caption (left-padded, with its emb_mask, like generate()): the
T-token prefix of ONE request is prefilled on a private two-row engine (CaptionEmbedder MLP + all T positions per layer at once,
the first image token sampled there), its K/V slots 0..T-1 and its mask rows are copied into the slot's rows, and the slot joins
the step batch at position T -- the per-request prefill of the vLLM fork (serve/model_runner.py:982-1076) without stalling the
other slots' graph replays for longer than that one prefill.

Slot-count buckets (`slot_buckets=(16, 32)` next to `slots=64`; the captured-batch-size ladder of serve/model_runner.py:36-41):
one engine + one captured graph per bucket, built on first use.  Before a step the batcher picks the smallest bucket that holds
the live requests (running + queued); growing is immediate, shrinking waits `shrink_after` steps.  A switch moves every running
request -- its K/V rows, token, position, class id and mask rows, and (when its slot index changes) its noise block and token row
-- into the new bucket's rows with stream-ordered device copies, so a lightly loaded server replays a 32-row graph instead of a
128-row one and still never recomputes a prefix.

Every request produces exactly the tokens a batch-of-one `generate()` would with the same noise: rows are independent
through every kernel (tests/test_gpu_serve.py holds that to the oracle, token for token, in fp32).
"""
from __future__ import annotations

import collections
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from .engine import DecodeEngine
from .gpt import find_multiple


def bucket_ladder(slots: int, slot_buckets: Optional[Sequence[int]]) -> List[int]:
    """Ascending captured slot counts; the full `slots` is always the last one."""
    if slots < 1:
        raise ValueError("slots must be >= 1")
    sizes = sorted({int(b) for b in (slot_buckets or ())})
    if sizes and (sizes[0] < 1 or sizes[-1] > slots):
        raise ValueError(f"slot_buckets must lie in [1, {slots}]")
    return [b for b in sizes if b < slots] + [slots]


def pick_bucket(sizes: Sequence[int], live: int) -> int:
    """Smallest captured slot count that holds `live` requests (the largest when none does)."""
    for b in sizes:
        if b >= live:
            return b
    return sizes[-1]


def plan_moves(active: Sequence[bool], new_slots: int):
    """(old slot, new slot) for every running request when the slot count becomes `new_slots`: a request keeps its index when
    that still exists, otherwise it takes the lowest free one."""
    run = [s for s, a in enumerate(active) if a]
    if len(run) > new_slots:
        raise ValueError(f"{len(run)} running requests do not fit {new_slots} slots")
    free = [t for t in range(new_slots) if not (t < len(active) and active[t])]
    return [(s, s) if s < new_slots else (s, free.pop(0)) for s in run]


class _Bucket:
    """One captured slot count: its engine (KV rows, activations), per-row positions / class ids and its step graph."""

    def __init__(self, model, slots: int, use_cfg: bool, S8: int, dtype, dev):
        self.B = slots
        self.B2 = 2 * slots if use_cfg else slots
        self.eng = e = DecodeEngine(model, self.B2, S8, dtype)
        R = e.MTs * 16
        # parked rows sit at position 1: a valid slot that is not 0 (= "fresh class-conditional request" for lgen_embed_rows) and
        # costs the attention two keys per idle row instead of a full sequence
        self.row_pos = torch.full((R,), 1, dtype=torch.int32, device=dev)
        self.cond = torch.full((R,), int(model.num_classes), dtype=torch.int32, device=dev)   # uncond rows: the null class
        e.pos_rows = self.row_pos  # switches the engine's layer chain to the per-row entry points
        self.graph = None


class ContinuousBatcher:
    def __init__(self, model, slots: int, max_new_tokens: int, cfg_scale: float = 1.0, cfg_interval: int = -1,
                 temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, sample_logits: bool = True,
                 slot_buckets: Optional[Sequence[int]] = None, shrink_after: int = 8):
        if model.model_type not in ("c2i", "t2i"):
            raise Exception("please check model type")
        dev = model.tok_embeddings.weight.device
        if dev.type != "cuda":
            raise RuntimeError("llamagen_amd.serve runs only on an AMD GPU through the HIP library (no CPU fallback)")
        self.model, self.dev = model, dev
        self.num_classes, self.V = int(model.num_classes), int(model.config.vocab_size)
        self.t2i = model.model_type == "t2i"
        self.T = int(model.cls_token_num) if self.t2i else 1
        self.slots, self.N = int(slots), max_new_tokens
        self.use_cfg = cfg_scale > 1.0
        self._dtype = model.tok_embeddings.weight.dtype
        self.S8 = find_multiple(self.T + max_new_tokens, 8)
        if max_new_tokens > model.block_size:
            raise IndexError(f"{max_new_tokens} tokens exceed block_size {model.block_size}")
        self.bucket_sizes = bucket_ladder(self.slots, slot_buckets)
        self.shrink_after = max(1, int(shrink_after))
        self._buckets: Dict[int, _Bucket] = {}
        self._shrink_wait = 0
        self.switches = 0
        self.cur = self._bucket(self.bucket_sizes[0] if len(self.bucket_sizes) > 1 else self.slots)
        # slot-indexed state shared by every bucket (bucket k uses slots 0 .. k-1)
        self.row_step = torch.full((slots,), max_new_tokens, dtype=torch.int32, device=dev)  # >= N: empty slot
        self.noise = torch.empty(slots, max_new_tokens, self.V, dtype=torch.float32, device=dev) if sample_logits else None
        self.seq = torch.zeros(slots, max_new_tokens, dtype=torch.int32, device=dev)
        self.sp = dict(cfg_scale=float(cfg_scale), cfg_interval=int(cfg_interval), temperature=float(temperature), top_k=int(top_k),
                       top_p=float(top_p), greedy=0 if sample_logits else 1)
        self._pe = None             # t2i: the private prefill engine (built on the first caption request)
        self._queue = collections.deque()
        self._slot_req: List[Optional[int]] = [None] * slots
        self._slot_left = [0] * slots
        self._next_id = 0
        self.steps_run = 0

    # the bucket in use: rows b and B + b pair up, so B is the CURRENT bucket's slot count
    B = property(lambda self: self.cur.B)
    B2 = property(lambda self: self.cur.B2)
    eng = property(lambda self: self.cur.eng)
    row_pos = property(lambda self: self.cur.row_pos)
    cond = property(lambda self: self.cur.cond)

    def _bucket(self, size: int) -> _Bucket:
        if size not in self._buckets:
            self._buckets[size] = _Bucket(self.model, size, self.use_cfg, self.S8, self._dtype, self.dev)
        return self._buckets[size]

    # ---- bucket choice and migration --------------------------------------------------------------------------------------
    def _live(self) -> int:
        return sum(r is not None for r in self._slot_req) + len(self._queue)

    def _choose(self):
        """Grow at once when the live requests do not fit, shrink after `shrink_after` consecutive steps in which they would."""
        want = pick_bucket(self.bucket_sizes, self._live())
        if want > self.cur.B:
            self._switch(self._bucket(want))
        elif want < self.cur.B:
            self._shrink_wait += 1
            if self._shrink_wait >= self.shrink_after:
                self._switch(self._bucket(want))
        else:
            self._shrink_wait = 0

    def _switch(self, nb: _Bucket):
        """Move every running request from the current bucket's rows into `nb`'s (stream-ordered device copies)."""
        ob = self.cur
        moves = plan_moves([r is not None for r in self._slot_req], nb.B)
        oe, ne = ob.eng, nb.eng
        for s, t in moves:
            for src, dst in ((s, t),) + (((ob.B + s, nb.B + t),) if self.use_cfg else ()):
                ne.k_cache[:, dst].copy_(oe.k_cache[:, src])
                ne.v_cache[:, dst].copy_(oe.v_cache[:, src])
                ne.cur_tok[dst:dst + 1].copy_(oe.cur_tok[src:src + 1])
                nb.row_pos[dst:dst + 1].copy_(ob.row_pos[src:src + 1])
                nb.cond[dst:dst + 1].copy_(ob.cond[src:src + 1])
                if self.t2i:
                    ne.causal_mask[dst].copy_(oe.causal_mask[src])
            if t != s:
                if self.noise is not None:
                    self.noise[t].copy_(self.noise[s])
                self.seq[t].copy_(self.seq[s])
                self.row_step[t:t + 1].copy_(self.row_step[s:s + 1])
                self.row_step[s] = self.N
                self._slot_req[t], self._slot_left[t] = self._slot_req[s], self._slot_left[s]
                self._slot_req[s], self._slot_left[s] = None, 0
        ob.row_pos.fill_(1)  # the old bucket's rows idle until it is used again
        self.cur = nb
        self._shrink_wait = 0
        self.switches += 1

    # ---- requests ---------------------------------------------------------------------------------------------------
    def submit(self, class_label, noise: Optional[torch.Tensor] = None, emb_mask: Optional[torch.Tensor] = None) -> int:
        """Queue one image request; `noise` (optional, [N, V] fp32 Exp(1) draws) replaces the default generator's draw.
        c2i: `class_label` is the class id.  t2i: `class_label` is the caption embedding [T, caption_dim] (left-padded and already
        multiplied by its mask, sample_t2i.py:95-108) and `emb_mask` its [T] validity mask."""
        if self.t2i:
            cap = class_label
            if not torch.is_tensor(cap) or cap.dim() != 2 or cap.shape[0] != self.T:
                raise ValueError(f"a text-conditional request is a [{self.T}, caption_dim] caption embedding")
            if emb_mask is not None and tuple(emb_mask.shape) != (self.T,):
                raise ValueError(f"emb_mask must be [{self.T}]")
            label = (cap, emb_mask)
        else:
            label = int(class_label)
            if not 0 <= label <= self.num_classes:  # num_classes itself is the null class (LabelEmbedder table row, gpt.py:66-72);
                raise IndexError(f"class label {label} outside [0, {self.num_classes}]")  # the reference's nn.Embedding raises too
        if noise is not None:
            if self.noise is None:
                raise ValueError("noise given but the batcher was built with sample_logits=False (greedy)")
            if tuple(noise.shape) != (self.N, self.V):
                raise ValueError(f"noise must be [{self.N}, {self.V}] fp32 Exp(1) draws, got {tuple(noise.shape)}")
        rid = self._next_id
        self._next_id += 1
        self._queue.append((rid, label, noise))
        return rid

    def _load(self, b: int, rid: int, label, noise):
        """Slot b <- a fresh request: position / step 0, its class label, its noise block (stream-ordered scalar writes)."""
        if self.noise is not None:
            if noise is not None:
                self.noise[b].copy_(noise.to(self.dev))
            else:
                self.noise[b].exponential_(1.0)
        if self.t2i:
            self._load_caption(b, *label)
            self._slot_req[b], self._slot_left[b] = rid, self.N - 1   # the first token was sampled by the prefill
            return
        self.row_pos[b] = 0
        self.row_step[b] = 0
        self.cond[b] = label
        if self.use_cfg:
            self.row_pos[self.B + b] = 0
        self._slot_req[b], self._slot_left[b] = rid, self.N

    def _load_caption(self, b: int, cap: torch.Tensor, emb_mask):
        """Per-request prefill of a text-conditional request (generate.py:128-170 for ONE image): caption (+ null caption with
        CFG) through the CaptionEmbedder and all T prefix positions of every layer on the private engine, first image token sampled
        from the slot's noise block, then K/V slots 0..T-1, mask rows, token and counters move into slot b of the step batch."""
        e, T, dev = self.eng, self.T, self.dev
        rows = 2 if self.use_cfg else 1
        if self._pe is None:
            self._pe = DecodeEngine(self.model, rows, self.S8, e.dtype)
        pe = self._pe
        pe.reset(rows)
        cond = cap.to(device=dev, dtype=e.dtype).unsqueeze(0)
        if self.use_cfg:
            cond = torch.cat([cond, self.model.cls_embedding.uncond_embedding.to(device=dev, dtype=e.dtype).unsqueeze(0)])
        if emb_mask is not None:  # generate.py:154-163 on the private engine's mask; both rows see the same pattern
            cm = pe.causal_mask
            cm[:, :, :T] = cm[:, :, :T] & emb_mask.to(dev).ne(0).view(1, 1, T)
            cm |= torch.eye(self.S8, dtype=torch.bool, device=dev)
        pe.use_mask = True
        pe._prefill_prefix(pe.caption_embed(cond))
        pe.state[0] = T - 1
        pe.state[1] = 0
        pe._final_logits()
        sp = dict(use_cfg=self.use_cfg, cfg_scale=self.sp["cfg_scale"], cfg_interval=self.sp["cfg_interval"],
                  temperature=self.sp["temperature"], top_k=self.sp["top_k"], top_p=self.sp["top_p"],
                  sample_logits=not self.sp["greedy"])
        pe.noise = self.noise[b].view(self.N, 1, self.V) if self.noise is not None else None
        pe._sample(1, sp)
        for src, dst in ((0, b),) + (((1, self.B + b),) if self.use_cfg else ()):
            e.k_cache[:, dst, :, :T].copy_(pe.k_cache[:, src, :, :T])
            e.v_cache[:, dst, :, :T].copy_(pe.v_cache[:, src, :, :T])
            e.causal_mask[dst].copy_(pe.causal_mask[src])
            e.cur_tok[dst:dst + 1].copy_(pe.cur_tok[0:1])
            self.row_pos[dst] = T
        self.seq[b, 0:1].copy_(pe.seq[0, 0:1])
        self.row_step[b] = 1

    # ---- one step of the whole slot batch -------------------------------------------------------------------------------
    def _step(self):
        e, lib, sp = self.eng, self.eng.lib, self.sp
        cls = e.tok_emb if self.t2i else e.cls_emb   # t2i rows never sit at position 0 (their prefix is prefilled): table unused
        L.check(lib.lgen_embed_rows(L.ptr(e.tok_emb), L.ptr(cls), L.ptr(e.cur_tok), L.ptr(self.cond), L.ptr(self.row_pos),
                                    L.ptr(e.hp), L.ptr(e.ssq) if e.fuse_norm else 0, self.B2, e.MTs, e.d, e.tok_emb.shape[0],
                                    cls.shape[0], e.dt, L.stream()), "embed_rows")
        e.ssq_parts = e.d // 16
        e._layers_and_logits()
        L.check(lib.lgen_sample_rows(L.ptr(e.logits), L.ptr(self.noise), L.ptr(e.cur_tok), L.ptr(self.seq), L.ptr(self.row_step),
                                     L.ptr(self.row_pos), self.N, self.B, e.V, self.N, 1 if self.use_cfg else 0, sp["cfg_scale"],
                                     sp["cfg_interval"], sp["temperature"], sp["top_k"], sp["top_p"], sp["greedy"], e.dt, L.stream()),
                "sample_rows")

    def run(self, use_graph: bool = True) -> Dict[int, torch.Tensor]:
        """Drain the queue; returns {request id: int32 [N] token ids} (device tensors, enqueued on the current stream)."""
        done: Dict[int, torch.Tensor] = {}
        with torch.no_grad():
            while self._queue or any(r is not None for r in self._slot_req):
                if len(self.bucket_sizes) > 1:
                    self._choose()
                for b in range(self.B):  # refill free slots (arrival order)
                    if self._slot_req[b] is None and self._queue:
                        self._load(b, *self._queue.popleft())
                if use_graph:
                    if self.cur.graph is None:
                        self._step()  # warm every kernel once, eagerly
                        self.steps_run += 1
                        self._account(done)
                        g = torch.cuda.CUDAGraph()
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g):
                            self._step()
                        self.cur.graph = g
                        continue
                    self.cur.graph.replay()
                else:
                    self._step()
                self.steps_run += 1
                self._account(done)
        return done

    def _account(self, done):
        for b in range(self.B):
            if self._slot_req[b] is None:
                continue
            self._slot_left[b] -= 1
            if self._slot_left[b] == 0:  # the step just enqueued wrote this request's last token
                done[self._slot_req[b]] = self.seq[b].clone()
                self._slot_req[b] = None
                self.row_pos[b] = 1          # park the rows again (after that step, stream-ordered): an idle row at its final
                if self.use_cfg:             # position would keep costing the attention a full sequence
                    self.row_pos[self.B + b] = 1
