"""f4 first slice: token-level continuous batching (llamagen_amd/serve.py) -- per-row positions through embed / RoPE / KV append /
attention / sampler, paired cond+uncond rows, slots refilled between graph replays.  The property that makes it correct: a
request's tokens do not depend on its neighbours, its slot or its arrival time -- each one must equal the oracle's batch-of-one
generate() on the same noise, token for token (fp32 storage)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import llamagen_oracle as O  # noqa: E402
from llamagen_amd.gpt import ModelArgs, Transformer  # noqa: E402
from llamagen_amd.testing import synth_for_module  # noqa: E402

KW = dict(n_layer=2, n_head=4, dim=256, vocab_size=1024, block_size=16, num_classes=10, cls_token_num=1, model_type="c2i")


def _model(dtype):
    m = Transformer(ModelArgs(**KW))
    sd = synth_for_module(m, seed=1, lin_std=0.05)
    m.load_state_dict(sd, strict=False)
    return m.to(device=torch.device("cuda:0"), dtype=dtype).eval(), sd


@pytest.mark.parametrize("cfg_scale,slots,use_graph", [(4.0, 3, True), (1.0, 2, True), (3.0, 5, False)])
def test_requests_are_independent_of_neighbours(cfg_scale, slots, use_graph):
    from llamagen_amd.serve import ContinuousBatcher
    m, sd = _model(torch.float32)
    N, V, nreq = 16, KW["vocab_size"], 8
    g = torch.Generator().manual_seed(11)
    labels = torch.randint(0, 10, (nreq,), generator=g).tolist()
    noises = [torch.empty(N, V).exponential_(1, generator=g) for _ in range(nreq)]
    skw = dict(cfg_scale=cfg_scale, cfg_interval=-1, temperature=1.0, top_k=100, top_p=1.0, sample_logits=True)
    cb = ContinuousBatcher(m, slots, N, **skw)
    ids = [cb.submit(l, n) for l, n in zip(labels, noises)]
    out = cb.run(use_graph=use_graph)
    torch.cuda.synchronize()
    assert sorted(out) == ids and cb.steps_run >= N * ((nreq + slots - 1) // slots)
    model = O.GPTOracle(O.GPTConfig(**KW), sd, torch.float32)
    for rid, label, nz in zip(ids, labels, noises):
        it = iter(nz)
        ref = O.generate(model, torch.tensor([label]), N, noise_fn=lambda shape: next(it).view(1, -1), **skw)
        np.testing.assert_array_equal(out[rid].cpu().numpy(), ref[0].numpy(), err_msg=f"request {rid}")


def test_staggered_arrivals_and_generate_agreement():
    """Requests submitted while others are mid-sequence (different positions in one step batch) still reproduce generate()
    of a batch of one on the same noise (fp32 storage, cfg_interval in play, hipGraph replay after hand-driven steps)."""
    from llamagen_amd import generate
    from llamagen_amd.serve import ContinuousBatcher
    m, _ = _model(torch.float32)
    N, V = 16, KW["vocab_size"]
    g = torch.Generator().manual_seed(5)
    skw = dict(cfg_scale=4.0, cfg_interval=3, temperature=0.9, top_k=50, top_p=1.0, sample_logits=True)
    cb = ContinuousBatcher(m, 2, N, **skw)
    noises = [torch.empty(N, V).exponential_(1, generator=g) for _ in range(5)]
    labels = [3, 7, 1, 9, 0]
    ids = [cb.submit(labels[0], noises[0])]
    out = {}
    # drive the loop by hand: one request alone for 5 steps, then the rest arrive
    cb._load(0, *cb._queue.popleft())
    for _ in range(5):
        cb._step()
        cb.steps_run += 1
        cb._account(out)
    ids += [cb.submit(l, n) for l, n in zip(labels[1:], noises[1:])]
    out.update(cb.run())
    torch.cuda.synchronize()
    m2, _ = _model(torch.float32)
    for rid, label, nz in zip(ids, labels, noises):
        ref = generate(m2, torch.tensor([label], device="cuda:0"), N, _noise_seq=nz.view(N, 1, V), **skw)
        assert torch.equal(out[rid].cpu(), ref[0].cpu()), rid


KW_T2I = dict(n_layer=2, n_head=4, dim=256, vocab_size=1024, block_size=16, cls_token_num=120, caption_dim=64, model_type="t2i")


@pytest.mark.parametrize("cfg_scale,slots", [(7.5, 2), (1.0, 3)])
def test_text_conditional_requests_with_staggered_arrivals(cfg_scale, slots):
    """Round 3: t2i requests (120-token caption prefix prefilled per request on a private engine, then joined to the slot batch at
    position T).  Requests arrive while others are mid-sequence; every one must equal the oracle's batch-of-one generate() with its
    caption, emb_mask and noise, token for token (fp32 storage)."""
    from llamagen_amd.serve import ContinuousBatcher
    m = Transformer(ModelArgs(**KW_T2I))
    sd = synth_for_module(m, seed=7, lin_std=0.05)
    m.load_state_dict(sd, strict=False)
    m = m.to(device=torch.device("cuda:0"), dtype=torch.float32).eval()
    N, V, T, C, nreq = 16, KW_T2I["vocab_size"], 120, 64, 5
    g = torch.Generator().manual_seed(21)
    caps, masks = [], []
    for _ in range(nreq):
        n = int(torch.randint(1, T + 1, (1,), generator=g))
        mk = torch.zeros(T, dtype=torch.int64)
        mk[T - n:] = 1                                     # left-padded: valid tokens at the end (sample_t2i.py:95-107)
        caps.append(torch.randn(T, C, generator=g) * mk[:, None])
        masks.append(mk)
    noises = [torch.empty(N, V).exponential_(1, generator=g) for _ in range(nreq)]
    skw = dict(cfg_scale=cfg_scale, cfg_interval=-1, temperature=1.0, top_k=100, top_p=1.0, sample_logits=True)
    cb = ContinuousBatcher(m, slots, N, **skw)
    ids = [cb.submit(caps[0], noises[0], masks[0])]
    out = {}
    cb._load(0, *cb._queue.popleft())                      # one request alone for 4 steps, then the rest arrive
    for _ in range(4):
        cb._step()
        cb.steps_run += 1
        cb._account(out)
    ids += [cb.submit(c, n, k) for c, n, k in zip(caps[1:], noises[1:], masks[1:])]
    out.update(cb.run())
    torch.cuda.synchronize()
    assert sorted(out) == ids
    model = O.GPTOracle(O.GPTConfig(**KW_T2I), sd, torch.float32)
    for rid, cap, mk, nz in zip(ids, caps, masks, noises):
        it = iter(nz)
        ref = O.generate(model, cap.unsqueeze(0), N, emb_masks=mk.unsqueeze(0), noise_fn=lambda shape: next(it).view(1, -1), **skw)
        np.testing.assert_array_equal(out[rid].cpu().numpy(), ref[0].numpy(), err_msg=f"request {rid}")
    with pytest.raises(ValueError):
        cb.submit(torch.zeros(T - 1, C))


@pytest.mark.parametrize("kind,cfg_scale", [("c2i", 4.0), ("c2i", 1.0), ("t2i", 7.5)])
def test_slot_count_buckets_move_running_requests_between_graphs(kind, cfg_scale):
    """Captured slot counts 1 / 2 / 6: one request starts alone in the 1-slot graph, a burst grows the batch mid-sequence, the tail
    shrinks it again (requests relocated to low slots mid-sequence).  Whatever graphs a request passed through, its tokens are the
    oracle's batch-of-one generate() on its noise (fp32 storage)."""
    from llamagen_amd.serve import ContinuousBatcher
    kw = KW if kind == "c2i" else KW_T2I
    m = Transformer(ModelArgs(**kw))
    sd = synth_for_module(m, seed=3, lin_std=0.05)
    m.load_state_dict(sd, strict=False)
    m = m.to(device=torch.device("cuda:0"), dtype=torch.float32).eval()
    N, V, nreq = 16, kw["vocab_size"], 9
    g = torch.Generator().manual_seed(31)
    if kind == "c2i":
        conds = [(int(torch.randint(0, 10, (1,), generator=g)), None) for _ in range(nreq)]
    else:
        T, C = 120, 64
        conds = []
        for _ in range(nreq):
            n = int(torch.randint(1, T + 1, (1,), generator=g))
            mk = torch.zeros(T, dtype=torch.int64)
            mk[T - n:] = 1
            conds.append((torch.randn(T, C, generator=g) * mk[:, None], mk))
    noises = [torch.empty(N, V).exponential_(1, generator=g) for _ in range(nreq)]
    skw = dict(cfg_scale=cfg_scale, cfg_interval=-1, temperature=1.0, top_k=100, top_p=1.0, sample_logits=True)
    cb = ContinuousBatcher(m, 6, N, slot_buckets=(1, 2), shrink_after=2, **skw)
    assert cb.bucket_sizes == [1, 2, 6] and cb.B == 1
    sub = lambda i: cb.submit(conds[i][0], noises[i], conds[i][1]) if kind == "t2i" else cb.submit(conds[i][0], noises[i])
    ids, out = [sub(0)], {}
    cb._load(0, *cb._queue.popleft())
    for _ in range(3):                                   # alone in the 1-slot bucket (eager steps)
        cb._step()
        cb.steps_run += 1
        cb._account(out)
    ids += [sub(i) for i in range(1, 6)]
    for _ in range(7):                                   # 6 live: the 6-slot bucket; stop mid-sequence
        cb._choose()
        for b in range(cb.B):
            if cb._slot_req[b] is None and cb._queue:
                cb._load(b, *cb._queue.popleft())
        cb._step()
        cb.steps_run += 1
        cb._account(out)
    assert cb.B == 6
    out.update(cb.run())                                 # graphs from here on; request 0 ends first, the rest together
    ids += [sub(i) for i in range(6, 8)]
    cb._load(4, *cb._queue.popleft())                    # placed in HIGH slots of the 6-slot bucket: the shrink to 2 slots two steps
    cb._load(5, *cb._queue.popleft())                    # later relocates them (K/V rows, noise block, token row) to slots 0 and 1
    out.update(cb.run())
    assert cb.B == 2
    ids.append(sub(8))
    out.update(cb.run())                                 # 1 live -> 1 slot
    torch.cuda.synchronize()
    assert sorted(out) == ids and cb.B == 1 and cb.switches >= 3
    model = O.GPTOracle(O.GPTConfig(**kw), sd, torch.float32)
    for rid, (c, mk), nz in zip(ids, conds, noises):
        it = iter(nz)
        if kind == "c2i":
            ref = O.generate(model, torch.tensor([c]), N, noise_fn=lambda shape: next(it).view(1, -1), **skw)
        else:
            ref = O.generate(model, c.unsqueeze(0), N, emb_masks=mk.unsqueeze(0), noise_fn=lambda shape: next(it).view(1, -1), **skw)
        np.testing.assert_array_equal(out[rid].cpu().numpy(), ref[0].numpy(), err_msg=f"request {rid}")
