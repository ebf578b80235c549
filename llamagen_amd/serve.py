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

Every request produces exactly the tokens a batch-of-one `generate()` would with the same noise: rows are independent
through every kernel (tests/test_gpu_serve.py holds that to the oracle, token for token, in fp32).
"""
from __future__ import annotations

import collections
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .engine import DecodeEngine
from .gpt import find_multiple


class ContinuousBatcher:
    def __init__(self, model, slots: int, max_new_tokens: int, cfg_scale: float = 1.0, cfg_interval: int = -1,
                 temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, sample_logits: bool = True):
        if model.model_type not in ("c2i", "t2i"):
            raise Exception("please check model type")
        dev = model.tok_embeddings.weight.device
        if dev.type != "cuda":
            raise RuntimeError("llamagen_amd.serve runs only on an AMD GPU through the HIP library (no CPU fallback)")
        self.model, self.dev = model, dev
        self.num_classes, self.V = int(model.num_classes), int(model.config.vocab_size)
        self.t2i = model.model_type == "t2i"
        self.T = int(model.cls_token_num) if self.t2i else 1
        self.B, self.N = slots, max_new_tokens
        self.use_cfg = cfg_scale > 1.0
        self.B2 = 2 * slots if self.use_cfg else slots
        dtype = model.tok_embeddings.weight.dtype
        self.S8 = find_multiple(self.T + max_new_tokens, 8)
        if max_new_tokens > model.block_size:
            raise IndexError(f"{max_new_tokens} tokens exceed block_size {model.block_size}")
        self.eng = DecodeEngine(model, self.B2, self.S8, dtype)
        e = self.eng
        R = e.MTs * 16
        # parked rows sit at position 1: a valid slot that is not 0 (= "fresh class-conditional request" for lgen_embed_rows) and
        # costs the attention two keys per idle row instead of a full sequence
        self.row_pos = torch.full((R,), 1, dtype=torch.int32, device=dev)
        self.row_step = torch.full((slots,), max_new_tokens, dtype=torch.int32, device=dev)  # >= N: empty slot
        self.cond = torch.full((R,), model.num_classes, dtype=torch.int32, device=dev)      # uncond rows: the null class
        self.noise = torch.empty(slots, max_new_tokens, e.V, dtype=torch.float32, device=dev) if sample_logits else None
        self.seq = torch.zeros(slots, max_new_tokens, dtype=torch.int32, device=dev)
        self.sp = dict(cfg_scale=float(cfg_scale), cfg_interval=int(cfg_interval), temperature=float(temperature), top_k=int(top_k),
                       top_p=float(top_p), greedy=0 if sample_logits else 1)
        e.pos_rows = self.row_pos  # switches the engine's layer chain to the per-row entry points
        self._pe = None             # t2i: the private prefill engine (built on the first caption request)
        self._graph = None
        self._queue = collections.deque()
        self._slot_req: List[Optional[int]] = [None] * slots
        self._slot_left = [0] * slots
        self._next_id = 0
        self.steps_run = 0

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
                for b in range(self.B):  # refill free slots (arrival order)
                    if self._slot_req[b] is None and self._queue:
                        self._load(b, *self._queue.popleft())
                if use_graph:
                    if self._graph is None:
                        self._step()  # warm every kernel once, eagerly
                        self.steps_run += 1
                        self._account(done)
                        g = torch.cuda.CUDAGraph()
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g):
                            self._step()
                        self._graph = g
                        continue
                    self._graph.replay()
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
