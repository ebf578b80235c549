#!/usr/bin/env python3
"""Headline benchmark: images/sec for LlamaGen-L 384x384 class-conditional sampling
(generate 576 tokens with CFG 4.0 / top-k 2000, then VQ decode_code) -- BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W            (N > 1: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config {2,3,4,5}                        (BASELINE.json configs[1..4]; default 2 = the headline)

One "step" = one pass of the hot path over one batch of 32 images per GPU (synthetic class labels,
random-init weights of the real architecture with output.weight re-randomised, bf16 GPT + fp32-class
VQ decoder), inputs resident in HBM.  Consecutive steps are independent batches, so `--batches-per-chain` of
them share one decode chain (their rows are concatenated; every image is what its own generate() call
would produce) and `--lanes` chains (default: 1..3, chosen from K) are kept in flight per GPU on separate HIP streams
(llamagen_amd/pipeline.py: the decode chain is latency-bound, chains interleave on the chip); all K timed steps start and
finish inside the timed region.  N > 1 shards independent images over ranks (weak scaling, no collective during
generation) and ends every step with ONE RCCL gather of the decoded batch to rank 0.  Rank 0 prints
one JSON line with `roofline` (dominant kernel = decode attention, measured live with HIP events on
its launch stream) and `cpu_baseline` (the CPU oracle / reference timed on a bounded sample).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import time

# HIP maps streams onto this many hardware queues (default 4); lanes that share a queue serialise
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json `configs` (index 1..4; configs[0] is the CPU-only case, timed by cpu_baseline()["c1"]).  `batch` is per GPU and
# per step; config 3 is "batch 256 sharded across 8 MI355X" = 32 per GPU.  Sampling settings: sample_c2i.py:117-122 / sample_t2i.py:150-152.
CONFIGS = {
    2: dict(name="LlamaGen-L 384px c2i", gpt="GPT-L", params="343M", img=384, batch=32, cfg=4.0, top_k=2000, model_type="c2i", T=1,
            vq_gflop=570.1),
    3: dict(name="LlamaGen-XXL 384px c2i", gpt="GPT-XXL", params="1.4B", img=384, batch=32, cfg=4.0, top_k=2000, model_type="c2i", T=1,
            vq_gflop=570.1),
    4: dict(name="LlamaGen-3B 384px c2i", gpt="GPT-3B", params="3.1B", img=384, batch=64, cfg=4.0, top_k=2000, model_type="c2i", T=1,
            vq_gflop=570.1),
    5: dict(name="LlamaGen-XL t2i stage2 512px", gpt="GPT-XL", params="775M", img=512, batch=16, cfg=7.5, top_k=1000, model_type="t2i",
            T=120, vq_gflop=1021.6),
}
GPT_DIMS = {"GPT-B": (12, 12, 768), "GPT-L": (24, 16, 1024), "GPT-XL": (36, 20, 1280), "GPT-XXL": (48, 24, 1536),
            "GPT-3B": (24, 32, 3200)}   # (layers, heads, dim) of the registry sizes (gpt.py:438-461): the KV budget needs no model
GPT_NAME, IMG, BATCH, CFG, TOPK = "GPT-L", 384, 32, 4.0, 2000   # config 2 (the headline)
LAT = IMG // 16
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)
CAPTION_DIM = 2048     # T5-XL feature width (gpt.py:40)


def plan_schedule(args, B, N, T):
    """Batches per decode chain and chains in flight (sets args.lanes) for this run, and the per-chain HBM need; raises SystemExit
    when the schedule does not fit the per-GPU budget -- BEFORE anything is allocated (needs no device and no model)."""
    # Schedule (round 4).  Config 2: TWO decode chains in flight, each carrying half of the run's steps (batches of 32) -- up to 12,
    # i.e. up to 768 rows with CFG.  Measured on MI355X (gpurun_out/ab2.log, attn_ab3/4.log; img/s, tile GEMMs + persistent
    # attention): 4 x 3 chains 107-110, 8 x 3 116, 8 x 2 119-120, 16 x 2 120, 16 x 1 107 -- per image the GEMM cost falls with the rows
    # of a chain (weights and launch chain amortised: 6.2 / 4.4 / 3.7 us per image-step at 256 / 512 / 1024 rows), attention and the
    # decoder do not care, and two wide chains overlap better than three narrow ones.  Every batch still is its own generate() (own
    # labels, own Exp(1) draws in the reference's order, rows never interact before the sampler).  Other configs: as in round 3.
    if args.batches_per_chain > 0:
        bpc = args.batches_per_chain
    elif args.config == 2:
        # chain widths an end-to-end oracle test names (tests/test_gpu_headline.py): up to 6 batches (<= 384 rows: skinny kernels
        # below 256 rows, TILE_SCHEDULES[16] from there) and 10 batches (640 rows, TILE_SCHEDULES[40]); 7..9 fall back to 6
        want = max(1, min(10, (args.steps + 1) // 2))
        bpc = want if (want <= 6 or want == 10) else 6
    elif args.config == 3:
        # GPT-XXL: two chains of up to EIGHT batches (512 rows, round 6: 46.4 img/s against 43.9 for two chains of six and 39.2 for
        # three chains of four, profiles/r06_widen.log); 2 x 8 x (10.6 GB of KV + 1.2 GB of noise) = 189 GB resident
        bpc = max(1, min(8, (args.steps + 1) // 2))
    elif args.config == 4:
        # GPT-3B: two chains of four batches of 64 (512 rows; round 6: 31.1 img/s against 30.0 for 2 x 2 with the 256-row shapes;
        # round 4 measured ONE chain of four at 25.2 against 28.2 for 2 x 2 -- two chains in flight matter more than the width)
        bpc = 4 if args.steps >= 8 else 2
    else:
        # GPT-XL t2i: two chains of eight batches of 16 (256 rows: tile family; round 6: 20.0 img/s against 17.8 for three chains of
        # four on the skinny kernels)
        bpc = 12 if args.steps >= 24 else (8 if args.steps >= 16 else 4)   # 2 x 12 (384 rows): 21.0-21.2 against 20.8 for 2 x 8
    chains = (args.steps + bpc - 1) // bpc
    n_layer, n_head, dim = GPT_DIMS[CONFIGS[args.config]["gpt"]]
    kvs = -(-(dim // n_head) // 8) * 8   # elements between key rows: head_dim rounded up to one 16-byte piece (engine.py)
    per_chain = (n_layer * 2 * B * bpc * n_head * (T + N + 8) * kvs * 2 * 2      # K and V slabs, CFG rows
                 + N * B * bpc * 16384 * 4)                                       # Exp(1) noise
    if args.lanes <= 0:
        if args.config in (2, 3):
            args.lanes = min(2, chains)
        else:
            # k chains in flight take ~T_k (round 3, 256-row chains with the decoder in 32-image pieces: 1, 1.71, 2.49 -- tools/exp_r3c.py);
            # a run of C chains on L lanes costs floor(C/L) * T_L + T_(C mod L): use the cheapest L
            Tk = {0: 0.0, 1: 1.0, 2: 1.44, 3: 2.02} if bpc == 1 else ({0: 0.0, 1: 1.0, 2: 1.5, 3: 2.1} if bpc < 4 else
                                                                       {0: 0.0, 1: 1.0, 2: 1.71, 3: 2.49})
            args.lanes = min((1, 2, 3), key=lambda l: (chains // l) * Tk[l] + Tk[chains % l])
        # KV slabs + noise of the chains in flight must fit the 288 GB of HBM3E with room for weights and decoder activations
        while args.lanes > 1 and args.lanes * per_chain > HBM_BUDGET_BYTES:
            args.lanes -= 1
    # per-rank HBM budget (the one-chain transparency leg needs one more chain's worth)
    if not args.no_one_chain and args.lanes > 1 and (args.lanes + 1) * per_chain > HBM_BUDGET_BYTES + 25e9 >= args.lanes * per_chain:
        args.no_one_chain = True   # the transparency leg would need a third chain's slabs: dropped for this schedule (the line then has no `images_per_s_with_one_chain_in_flight`)
    need = (args.lanes + (0 if (args.no_one_chain or args.lanes == 1) else 1)) * per_chain
    if need > HBM_BUDGET_BYTES + 25e9:
        raise SystemExit(f"schedule needs {need / 1e9:.0f} GB of KV slabs + noise per GPU ({args.lanes} chains x {bpc} batches of {B}): "
                         f"over the {HBM_BUDGET_BYTES / 1e9:.0f} GB budget; lower --batches-per-chain or --lanes")
    return bpc, per_chain


def build_models(dev, weight_seed=0, config=2):
    """Every rank builds the SAME model (the reference's DDP replicas load one checkpoint); only labels and
    sampling noise are per-rank (sample_c2i_ddp.py:47), seeded by the caller afterwards."""
    from llamagen_amd import GPT_models, VQ_models
    c = CONFIGS[config]
    torch.manual_seed(weight_seed)
    gpt = GPT_models[c["gpt"]](vocab_size=16384, block_size=(c["img"] // 16) ** 2, num_classes=1000, cls_token_num=c["T"],
                               model_type=c["model_type"])
    torch.nn.init.normal_(gpt.output.weight, 0, 0.02)  # zero-initialised in the reference (gpt.py:305)
    gpt = gpt.to(device=dev, dtype=torch.bfloat16).eval()
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
    return gpt, vq


def attention_bytes_per_generate(cfg, B2, N, T=1):
    """KV bytes the decode attention must read, summed over the N launches per
    layer (kv_len = pos+1 keys, K and V, bf16) -> bytes, launches."""
    hd = cfg.dim // cfg.n_head
    per_key = 2 * cfg.n_head * hd * 2  # K+V, all heads, bf16
    keys = sum(T + i for i in range(N))  # positions T-1 .. T+N-2 attend to pos+1 keys
    return per_key * B2 * keys * cfg.n_layer, N * cfg.n_layer


def measure_attention(gpt, B2, N, T=1, npos=12, reps=5):
    """Live HIP-event timing of the dominant kernel on the stream it is launched on: for `npos` cache
    positions spread over the sequence (T-1 .. T+N-2: the N launches per layer of one generate()), a captured chain of
    the L per-layer attention launches (each on its own layer's KV slab, so nothing is cache-resident) is replayed
    `reps` times between two events.
    Returns (seconds all N*L launches of one generate() take -- trapezoid over positions --, launches, per-position us)."""
    from llamagen_amd import _lib as L
    e = gpt._engine
    lib = e.lib
    e.k_cache.normal_(0, 1)  # random cache contents: zero-filled operands clock higher (MI355X guide, DVFS)
    e.v_cache.normal_(0, 1)
    stream = torch.cuda.Stream()
    pts = sorted(set([0] + [int(round(i * (N - 1) / (npos - 1))) for i in range(npos)]))  # token index 0..N-1 -> position T-1+i
    us = []
    with torch.cuda.stream(stream):
        for p in pts:
            e.state.copy_(torch.tensor([T - 1 + p, p], dtype=torch.int32, device=e.dev))

            def chain():
                for i in range(e.L):
                    L.check(lib.lgen_attn_decode(L.ptr(e.qbuf), L.ptr(e.k_cache[i]), L.ptr(e.v_cache[i]), L.ptr(e.ap),
                                                 L.ptr(e.state), 0, 0, B2, e.MTs, e.H, e.hd, e.hdp, e.S8, e.kvs, e.dt, e.attn_variant, L.stream()), "attn")
            chain()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                chain()
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    g.replay()
                e1.record(stream)
                stream.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * e.L))
            us.append(best)
    total_us = 0.0  # integrate the per-launch duration over token indices 0..N-1
    for (p0, t0), (p1, t1) in zip(zip(pts, us), zip(pts[1:], us[1:])):
        total_us += 0.5 * (t0 + t1) * (p1 - p0)
    total_us += us[-1]
    return total_us * e.L * 1e-6, N * e.L, dict(zip([T - 1 + p for p in pts], [round(u, 2) for u in us]))


def measure_gemms(gpt, reps=5):
    """Second kernel family of the decode step (the five GEMMs of a layer + lm_head): live HIP-event timing of captured chains that
    contain ONLY one kind of GEMM, over all layers' weights in turn (nothing cache-resident), through the engine's own launch
    methods, i.e. with the kernel family (big-M tile / skinny), tile shapes and schedules the decode graph uses.
    Returns {kind: (us per launch, weight bytes per launch)} + "_schedule"."""
    from llamagen_amd import _lib as L
    e = gpt._engine
    d, F, V = e.d, e.F, e.V
    e.ssq_parts = d // 16
    e.state.zero_()
    nw = lambda w: w if e.fuse_norm else None
    kinds = {
        "wqkv": (lambda: [e.qkv_gemm(0, w, e.hp, nw(w["an"])) for w in e.layers], 3 * d * d),
        "wo": (lambda: [e.gemm_kind("wo", w) for w in e.layers], d * d),
        "w13": (lambda: [e.gemm_kind("w13", w, e.hp, nw(w["fn"])) for w in e.layers], 2 * F * d),
        "w2": (lambda: [e.gemm_kind("w2", w) for w in e.layers], F * d),
        "lm_head": (lambda: [e.gemm_kind("head", None, e.hp, nw(e.norm_w)) for _ in range(4)], V * d),
    }
    esz = 2 if e.dtype == torch.bfloat16 else 4
    stream = torch.cuda.Stream()
    out = {}
    with torch.cuda.stream(stream):
        for kind, (fn, nparam) in kinds.items():
            launches = 4 if kind == "lm_head" else e.L
            fn()
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                fn()
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    g.replay()
                e1.record(stream)
                stream.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * launches))
            out[kind] = (best, nparam * esz)
    out["_schedule"] = e.gemm_schedule()
    return out


def gpt_phase_bytes(cfg, B, N, T=1, cfg_rows=True, esz=2):
    """SURVEY.md section 8d closed form: algorithmic HBM bytes of ONE generate() call of B images (B2 = 2B rows with CFG):
    N weight streams + KV reads + KV writes + logits + noise.  Returns a dict of the terms (bytes)."""
    B2 = 2 * B if cfg_rows else B
    d, L, H, V = cfg.dim, cfg.n_layer, cfg.n_head, cfg.vocab_size
    hd = d // H
    F = cfg.ffn_dim if hasattr(cfg, "ffn_dim") else None
    if F is None:  # llamagen_amd ModelArgs: same rule as gpt.py:151-159
        hidden = int(2 * 4 * d / 3)
        if getattr(cfg, "ffn_dim_multiplier", None) is not None:
            hidden = int(cfg.ffn_dim_multiplier * hidden)
        m = getattr(cfg, "multiple_of", 256)
        F = m * ((hidden + m - 1) // m)
    W = (L * (4 * d * d + 3 * d * F + 2 * d) + d * V + d) * esz      # weights streamed per step
    kvb = L * 2 * H * hd * esz                                         # KV bytes per token per row
    reads = kvb * B2 * (T + (N - 1) * (T + 1) + (N - 2) * (N - 1) // 2)
    writes = kvb * B2 * (T + N - 1)
    return {"weights": N * W, "kv_reads": reads, "kv_writes": writes, "logits": N * B2 * V * esz, "noise": N * B * V * 4,
            "weights_per_step": W, "kv_bytes_per_token": kvb}


def _cpu_decode_fns():
    """(kind, step, vq_decode) for the CPU leg: the reference's own modules when /root/reference is importable (build
    container), else the oracle port (the GPU box has no reference mount).  Both run GPT-L bf16 at B = 32 (CFG rows 64)."""
    N = LAT * LAT
    try:
        if not os.path.isdir("/root/reference") or os.environ.get("LGEN_BENCH_CPU_PORT") == "1":
            raise ImportError("no reference mount")
        sys.path.insert(0, "/root/reference")
        from autoregressive.models.gpt import GPT_models as RG
        from autoregressive.models.generate import decode_one_token
        from tokenizer.tokenizer_image.vq_model import VQ_models as RV
        torch.manual_seed(0)
        m = RG[GPT_NAME](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
        torch.nn.init.normal_(m.output.weight, 0, 0.02)
        m = m.to(torch.bfloat16).eval()
        with torch.device("cpu"):
            m.setup_caches(max_batch_size=2 * BATCH, max_seq_length=1 + N, dtype=torch.bfloat16)
        for blk in m.layers:
            blk.attention.kv_cache.k_cache.normal_(0, 1)
            blk.attention.kv_cache.v_cache.normal_(0, 1)
        tok = torch.randint(0, 16384, (BATCH, 1))

        def step(pos):
            with torch.no_grad():
                decode_one_token(m, tok, torch.tensor([pos]), CFG, True, temperature=1.0, top_k=TOPK, top_p=1.0, sample_logits=True)
        vq = RV["VQ-16"](codebook_size=16384, codebook_embed_dim=8).eval()

        def vq_decode(codes, shape):
            with torch.no_grad():
                return vq.decode_code(codes, shape)
        return "reference", step, vq_decode
    except Exception:  # noqa: BLE001 -- any import / construction problem: use the port
        pass
    from llamagen_amd import GPT_models, VQ_models
    from oracle import llamagen_oracle as O
    torch.manual_seed(0)
    m = GPT_models[GPT_NAME](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    cfg = O.GPTConfig(**O.GPT_SIZES[GPT_NAME], vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1)
    model = O.GPTOracle(cfg, dict(m.state_dict()), torch.bfloat16)
    model.setup_caches(2 * BATCH, 1 + N)
    for kc, vc in zip(model.k_cache, model.v_cache):  # plausible cache contents for the late steps
        kc.normal_(0, 1)
        vc.normal_(0, 1)
    tok = torch.randint(0, 16384, (BATCH, 1))

    def step(pos):
        lg = model.forward(torch.cat([tok, tok]), None, torch.tensor([pos]))
        O.sample(O.cfg_mix(lg, CFG)[:, -1], top_k=TOPK)
    vsd = dict(VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).state_dict())
    return "port", step, (lambda codes, shape: O.vq_decode_code(vsd, codes, shape))


def cpu_config1(threads):
    """BASELINE.json configs[0], timed IN FULL: LlamaGen-B 256 px (16 x 16 tokens), one image, cfg 1.0, fp32, top-k 2000, the
    reference's sample_c2i.py plumbing (generate + decode_code) on the host cores -- reference modules when importable, else the
    oracle port.  Returns {"value" (images/s), "seconds", "kind", "threads"}."""
    N, name = 256, "GPT-B"
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    kind = "port"
    try:
        if not os.path.isdir("/root/reference") or os.environ.get("LGEN_BENCH_CPU_PORT") == "1":
            raise ImportError("no reference mount")
        sys.path.insert(0, "/root/reference")
        from autoregressive.models.gpt import GPT_models as RG
        from autoregressive.models.generate import generate as rgen
        from tokenizer.tokenizer_image.vq_model import VQ_models as RV
        m = RG[name](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i").eval()
        torch.nn.init.normal_(m.output.weight, 0, 0.02)
        vq = RV["VQ-16"](codebook_size=16384, codebook_embed_dim=8).eval()

        def run():
            with torch.no_grad():
                idx = rgen(m, torch.tensor([207]), N, cfg_scale=1.0, cfg_interval=-1, temperature=1.0, top_k=TOPK, top_p=1.0,
                           sample_logits=True)
                return vq.decode_code(idx, [1, 8, 16, 16])
        kind = "reference"
    except Exception:  # noqa: BLE001
        from llamagen_amd import GPT_models, VQ_models
        from oracle import llamagen_oracle as O
        m = GPT_models[name](vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1, model_type="c2i")
        torch.nn.init.normal_(m.output.weight, 0, 0.02)
        cfg = O.GPTConfig(**O.GPT_SIZES[name], vocab_size=16384, block_size=N, num_classes=1000, cls_token_num=1)
        model = O.GPTOracle(cfg, dict(m.state_dict()), torch.float32)
        vsd = dict(VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8).state_dict())

        def run():
            idx = O.generate(model, torch.tensor([207]), N, cfg_scale=1.0, cfg_interval=-1, temperature=1.0, top_k=TOPK, top_p=1.0,
                             sample_logits=True)
            return O.vq_decode_code(vsd, idx, [1, 8, 16, 16])
    run()  # warm-up (allocator, thread pool, first-touch of the weights)
    secs = []
    for _ in range(2):
        t0 = time.time()
        img = run()
        secs.append(time.time() - t0)
    sec = min(secs)
    assert tuple(img.shape) == (1, 3, 256, 256)
    return {"value": round(1.0 / sec, 4), "unit": "images/s", "seconds": round(sec, 2), "seconds_all": [round(x, 2) for x in secs],
            "kind": kind, "threads": threads,
            "workload": "LlamaGen-B 256px c2i, 1 image, cfg 1.0, fp32, top-k 2000, generate + decode_code, timed in full (min of 2 after a warm-up run)"}


def cpu_baseline(steps=16, budget_s=30.0, with_c1=True):
    """CPU leg beside the GPU number (SURVEY 8d / BASELINE.md 4): the same workload -- GPT-L 384 px, B = 32, cfg 4.0 (64
    rows), top-k 2000, bf16 -- on the host cores, as a BOUNDED 32-step slice: `steps` decode steps at the start of the
    sequence and `steps` at its end (kv_len ~ 576), plus the VQ decode of 2 images; images/s = 32 / (mean step time x
    576 + 32 x VQ time per image).  Linear extrapolation, labelled as such.  The thread count is calibrated first, MIN-OF-3 per
    candidate, climbing 8, 16, 32, 64, 128, os.cpu_count() until a candidate's first step is > 2.5x slower than the best so far
    (torch's intra-op pool thrashes on a big host: one thread per logical core costs minutes per step on the 256-core GPU box),
    and reported next to the core count; `impl` names what ran.  `c1` = BASELINE configs[0] timed in full."""
    ncpu = os.cpu_count() or 1
    kind, step, vq_decode = _cpu_decode_fns()
    N = LAT * LAT
    best_thr, best_t, sweep = 1, 1e30, {}
    for thr in sorted({t for t in (8, 16, 32, 64, 128, ncpu) if t <= ncpu}) or [ncpu]:
        torch.set_num_threads(thr)
        t0 = time.time()
        step(N // 2)  # warm-up at this thread count
        warm = time.time() - t0
        if warm > 2.5 * best_t:
            # past the knee: torch's intra-op pool thrashes with more threads than the GEMMs can use (measured on the 256-logical-
            # core GPU box: 157 ms per step at 16 threads, 593 ms at 64, 193 s at 256).  Record the one sample and stop climbing.
            sweep[thr] = round(warm * 1e3, 1)
            break
        t = 1e30
        for _ in range(3):
            t0 = time.time()
            step(N // 2)
            t = min(t, time.time() - t0)
        sweep[thr] = round(t * 1e3, 1)
        if t < best_t:
            best_thr, best_t = thr, t
    torch.set_num_threads(best_thr)
    steps = max(2, min(steps, int(budget_s / 2 / max(best_t, 1e-3))))  # keep the whole leg near `budget_s`
    t0 = time.time()
    for i in range(steps):
        step(1 + i)
    t_early = (time.time() - t0) / steps
    t0 = time.time()
    for i in range(steps):
        step(N - steps + i)
    t_late = (time.time() - t0) / steps
    codes = torch.randint(0, 16384, (2, N))
    vq_decode(codes[:1, :16], [1, 8, 4, 4])  # warm-up
    t0 = time.time()
    vq_decode(codes, [2, 8, LAT, LAT])
    t_vq = (time.time() - t0) / 2
    per_batch = 0.5 * (t_early + t_late) * N + BATCH * t_vq
    impl = "reference modules (/root/reference, torch CPU)" if kind == "reference" else "oracle port (oracle/llamagen_oracle.py; no /root/reference on this box)"
    out = {"value": round(BATCH / per_batch, 5), "unit": "images/s (B=32 slice, extrapolated)",
           # `cores` is the contract's name for the THREADS the leg used (calibrated below), not the size of the host
           "cores": best_thr, "threads": best_thr, "host_cores": ncpu, "logical_cores": ncpu, "kind": kind, "impl": impl, "thread_sweep_ms_per_step_min_of_3": sweep,
           "sample": f"{impl}, GPT-L 384px "
                     f"bf16 at B=32 (64 CFG rows), {best_thr} threads of {ncpu} logical cores: {steps} early + {steps} late "
                     f"decode steps ({t_early*1e3:.0f} / {t_late*1e3:.0f} ms per step) extrapolated linearly to 576 tokens, "
                     f"+ 32 x the VQ decode time of one 384 px image ({t_vq:.2f} s, measured on 2)"}
    if with_c1:
        out["c1"] = cpu_config1(best_thr)
    if kind == "port":
        # same-host calibration of the port against the reference's own modules (tools/cpu_calibrate.py, run in the build container
        # where /root/reference exists): how much faster / slower the port is than the reference on the same slice
        try:
            cal = json.load(open(os.path.join(ROOT, "profiles", CPU_CAL_JSON)))
            out["port_over_reference"] = {"slice_images_per_s": cal["port_over_reference"]["slice"], "c1_images_per_s": cal["port_over_reference"]["c1"],
                                          "source": f"profiles/{CPU_CAL_JSON} (both run back to back in the build container, same threads)"}
            out["c1"]["port_over_reference"] = cal["port_over_reference"]["c1"]
        except Exception:  # noqa: BLE001
            out["port_over_reference"] = None
    return out


# ---- live HBM-traffic counters (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md "HBM") ----------
def live_traffic(rows, timeout_s=150):
    """Runs tools/pmc_attn_target.py (the decode attention on full-size KV slabs at three cache positions, one launch per
    layer) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a SEPARATE pass, `--pmc WRITE_SIZE`, and returns
    {"fetch_over_algorithmic", "write_bytes_per_launch", "source"} -- or None when rocprofv3 is not on PATH, a pass fails or
    times out (the caller then quotes the committed PMC pass instead).  FETCH_SIZE is KB and counts HALF of a wide coalesced
    read stream on gfx950: bytes = KB x 1024 x 2; WRITE_SIZE bytes = KB x 1024."""
    import csv, glob, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    tgt = os.path.join(ROOT, "tools", "pmc_attn_target.py")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                env = dict(os.environ, TMPDIR="/tmp", LGEN_PMC_ROWS=str(rows))
                r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", td, "--", sys.executable, tgt],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                if r.returncode != 0:
                    return None
                seq = []
                for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] == counter and "attn_decode" in row["Kernel_Name"]:   # attn_decode_kernel / attn_decode_persist_kernel
                            seq.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
                seq.sort()
                vals[counter] = [v for _, v in seq]
        return vals
    except Exception:  # noqa: BLE001 -- timeout, missing csv, ...: fall back to the committed pass
        return None


def other_configs(timeout_s=420):
    """BASELINE configs 3 (its per-GPU shape: GPT-XXL, batch 32 of the 256 over 8 GPUs), 4 and 5 next to the headline: `bench.py
    --config N` (the schedule of profiles/r0x_bench_configN.json, fewer steps) as a child process each, parsed from its JSON line.  A tested GEMM schedule is required there too (the child refuses an
    untested one before timing).  Returns {"config3": {...}, "config4": {...}, "config5": {...}}; a failed pass is reported as such,
    never invented."""
    out = {}
    for c, steps, warm in ((3, 16, 4), (4, 8, 2), (5, 24, 4)):
        cmd = [sys.executable, os.path.abspath(__file__), "--config", str(c), "--steps", str(steps), "--warmup", str(warm),
               "--no-cpu-baseline", "--no-live-traffic", "--no-solo", "--no-one-chain", "--no-roofline"]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1]) if (r.returncode == 0 and line) else None
        except Exception as ex:  # noqa: BLE001 -- timeout, bad JSON
            d, r = None, None
            err = repr(ex)
        if d is None:
            out[f"config{c}"] = {"error": (r.stderr[-300:] if r is not None else err), "seconds": round(time.time() - t0, 1)}
            continue
        out[f"config{c}"] = {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
                             "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
                             "batches_per_chain": d["config"]["batches_per_chain"],
                             "chains_in_flight_per_gpu": d["config"]["chains_in_flight_per_gpu"],
                             "gemm_schedule": d.get("gemm_schedule"), "seconds": round(time.time() - t0, 1),
                             "command": " ".join(cmd[1:])}
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line under torch.distributed.run, one rank per GPU
    (rank 0's JSON line passes through on stdout), and return its exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def standin_main(args, rank, local, world, rccl_ranks):
    """--standin (tests/test_bench_host.py): the rank plumbing of this file -- barrier + timed region, max over ranks, ONE gather
    per step, the JSON line -- with a CPU stand-in for the device step (gloo).  NOT a measurement and labelled as such."""
    import torch.distributed as dist
    from llamagen_amd import dist as ldist
    torch.manual_seed(ldist.rank_seed(0, rank, world))

    def step():
        time.sleep(0.002)
        return ldist.gather_to_root(torch.randint(0, 256, (BATCH, 8, 8, 3), dtype=torch.uint8))
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        assert out.shape == (BATCH * world, 8, 8, 3)
        print(json.dumps({"metric": "images/sec (whole node), LlamaGen-L 384px c2i", "value": round(BATCH * world * args.steps / dt, 3),
                          "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "rccl_ranks": rccl_ranks,
                          "data": "STAND-IN device step on CPU/gloo (launch-path test, NOT a measurement)",
                          "config": {"workload": "stand-in", "parallelism": f"dp{world}", "global_batch": BATCH * world}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[1..4]; 2 = the headline")
    ap.add_argument("--lanes", type=int, default=0, help="decode chains in flight per GPU (llamagen_amd/pipeline.py); "
                                                         "0 = pick 1..3 from the step count")
    ap.add_argument("--batches-per-chain", type=int, default=0, help="consecutive steps (batches) that share one decode chain; "
                                                                     "0 = the config's default")
    ap.add_argument("--steps-per-turn", type=int, default=1, help="decode steps a lane enqueues per scheduler turn")
    ap.add_argument("--vq-own-stream", action="store_true", help="decode images on a separate shared stream (measured slower)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="quote the committed PMC pass instead of running rocprofv3 --pmc now")
    ap.add_argument("--no-solo", action="store_true", help="skip the extra one-generate()-at-a-time leg (keeps a "
                                                           "rocprofv3 kernel average of this run to the timed chain shape)")
    ap.add_argument("--budget-check-only", action="store_true", help="plan the schedule, check the per-GPU HBM budget, print it, exit")
    ap.add_argument("--allow-untested-schedule", action="store_true", help="run a GEMM schedule no end-to-end parity test names")
    ap.add_argument("--no-one-chain", action="store_true", help="skip the extra one-chain-in-flight leg (it needs another chain's KV slabs)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short passes of BASELINE configs 4 and 5 (default config-2 run)")
    ap.add_argument("--standin", action="store_true", help=argparse.SUPPRESS)  # CPU/gloo test of the launch path, see standin_main
    args = ap.parse_args()

    if args.budget_check_only:   # no device, no model: the plan and its HBM need
        C0 = CONFIGS[args.config]
        bpc0, per0 = plan_schedule(args, C0["batch"], (C0["img"] // 16) ** 2, C0["T"])
        print(json.dumps({"config": args.config, "batches_per_chain": bpc0, "chains_in_flight_per_gpu": args.lanes,
                          "kv_plus_noise_GB_per_chain": round(per0 / 1e9, 1), "budget_GB": HBM_BUDGET_BYTES / 1e9}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # plain `python bench.py --gpus N`: become the launcher
        if not args.standin and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible")
        raise SystemExit(self_launch(args.gpus))

    from llamagen_amd import dist as ldist
    import torch.distributed as dist

    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # checked BEFORE the rendezvous (a short world would wait forever)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (or without a launcher: bench.py spawns its ranks itself)")
    rank, local, world = ldist.init_from_env("gloo" if args.standin else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or without a launcher: bench.py spawns its ranks itself)")
    rccl_ranks = 1
    if world > 1:  # one all-reduce before anything is timed: the ranks that really share the communicator
        one = torch.ones(1, device="cpu" if args.standin else f"cuda:{local}")
        dist.all_reduce(one)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size() == world
    if args.standin:
        return standin_main(args, rank, local, world, rccl_ranks)

    from llamagen_amd.pipeline import SamplingPipeline
    C = CONFIGS[args.config]
    B, img, cfg_scale, top_k, T = C["batch"], C["img"], C["cfg"], C["top_k"], C["T"]
    lat = img // 16
    N = lat * lat
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    gpt, vq = build_models(dev, 0, args.config)
    torch.manual_seed(ldist.rank_seed(0, rank, world))  # per-rank labels / sampling noise only
    skw = dict(cfg_scale=cfg_scale, cfg_interval=-1, temperature=1.0, top_k=top_k, top_p=1.0, sample_logits=True)
    t2i = C["model_type"] == "t2i"
    if t2i:  # sample_t2i.py:89-116: caption features left-padded to T with emb_masks; synthetic N(0,1) features, random lengths
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        lens = torch.randint(5, T + 1, (B,), generator=g)
        emb_masks = (torch.arange(T).unsqueeze(0) >= (T - lens).unsqueeze(1)).to(torch.int64).to(dev)
        skw["emb_masks"] = emb_masks
    bpc, per_chain = plan_schedule(args, B, N, T)
    pipe = SamplingPipeline(gpt, vq, lanes=args.lanes, steps_per_turn=args.steps_per_turn, vq_low_priority=args.vq_own_stream,
                            batches_per_chain=bpc,
                            vq_chunk=B if (bpc > 1 and args.lanes > 1) else 0)  # decode_code() batch by batch: finer interleaving
    pipe.prepare(B, N, **skw)  # setup (like loading weights): KV slabs, workspaces, decode graphs per lane
    torch.cuda.synchronize()
    # The GEMM schedule the timed region is about to replay must be one an end-to-end oracle test names (tests/test_gpu_headline.py:
    # the skinny kernels below 256 rows, a tested key of the model's pinned tile table from there).  Checked HERE, before anything is
    # timed and on every rank alike (round 4 checked after the timed region, on rank 0 only: ADVICE r4).
    eng0 = pipe.lanes[0].gpt._engine
    gsched0, sched_tested, sched_source = eng0.gemm_schedule(), bool(eng0.tile_schedule_tested()), eng0.tile_schedule_source()
    del eng0   # only the derived values stay: a live reference would keep lane 0's KV slabs + noise resident through the solo leg and other_configs()
    if not sched_tested and not args.allow_untested_schedule:
        raise SystemExit(f"GEMM schedule {gsched0} (source: {sched_source}) is not one tests/test_gpu_headline.py holds to the oracle "
                         "(--allow-untested-schedule to run it anyway)")

    from llamagen_amd.postprocess import to_uint8_hwc

    def finish(job_id, idx, im):
        """Per batch, enqueued on its lane's stream as soon as decode_code() is: fp32 [-1, 1] -> uint8 HWC (what the
        reference writes out, sample_c2i_ddp.py:143) and the step's ONE collective (uint8 gather to rank 0)."""
        return ldist.gather_to_root(to_uint8_hwc(im))

    def make_cond():
        if t2i:
            return torch.randn(B, T, CAPTION_DIM, device=dev).to(torch.bfloat16)
        return torch.randint(0, 1000, (B,), device=dev)

    def run_steps(k):
        """k steps; step = one batch of B images through generate() + decode_code() + uint8 conversion
        [+ gather]; consecutive steps ride on alternating lanes."""
        return pipe.run([make_cond() for _ in range(k)], N, on_done=finish, **skw)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    outs = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # transparency: the same workload with ONE chain in flight (2 chains, one after the other, on a lane of their own with the
    # single-chain GEMM shapes), outside the timed region above
    chain1 = None
    if args.lanes > 1 and not args.no_one_chain:
        view = gpt.lane_view()
        one = SamplingPipeline(view, vq, lanes=1, batches_per_chain=bpc)
        one.prepare(B, N, **skw)
        fence()
        t1 = time.perf_counter()
        one.run([make_cond() for _ in range(2 * bpc)], N, **skw)
        fence()
        chain1 = B * world * 2 * bpc / (time.perf_counter() - t1)
        del one, view
        torch.cuda.empty_cache()

    if rank == 0:
        out = outs[-1]
        assert out is not None and out.shape[0] == B * world and out.dtype == torch.uint8 and tuple(out.shape[1:]) == (img, img, 3)
        assert 8 < float(out.float().std()) < 128  # images, not a constant
        value = B * world * args.steps / dt
        if chain1 is None:
            chain1 = value
        rows = 2 * B * bpc
        phase = gpt_phase_bytes(gpt.config, B, N, T)
        res = {"metric": f"images/sec (whole node), {C['name']}", "value": round(value, 3), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "rccl_ranks": rccl_ranks,
               "config": {"workload": f"{C['name']} ({C['gpt']} {C['params']}): generate {N} tokens (cfg {cfg_scale}, top-k {top_k}, "
                                      f"bf16) + VQ-16 decode_code (fp32-class), batch {B} per step per GPU, random-init "
                                      f"weights; {bpc} consecutive steps share one decode chain ({rows} rows with CFG), "
                                      f"{args.lanes} chains in flight per GPU on separate HIP streams",
                          "baseline_config_index": args.config - 1,
                          "global_batch": B * world, "tokens_per_image": N, "parallelism": f"dp{world}",
                          "batches_per_chain": bpc, "chains_in_flight_per_gpu": args.lanes,
                          "steps_in_flight_per_gpu": args.lanes * bpc},
               # SURVEY 8d closed form for ONE generate() of this config's batch (per GPU)
               "algorithmic_bytes_per_generate": {k: int(v) for k, v in phase.items()},
               # the same workload with ONE chain (bpc batches) in flight at a time (no cross-chain overlap)
               "images_per_s_with_one_chain_in_flight": round(chain1, 3),
               "gemm_schedule": {"schedule": gsched0, "source": sched_source, "tested_end_to_end": sched_tested}}
        total_b = sum(phase[k] for k in ("weights", "kv_reads", "kv_writes", "logits", "noise"))
        res["gpt_phase_floor_ms_per_step_at_8TBps"] = round(total_b / 8e12 * 1e3, 1)
        pmc = {}
        try:  # PMC passes of THIS round (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs; tools/pmc_summary.py)
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_JSON if os.path.exists(os.path.join(ROOT, "profiles", PMC_JSON)) else "r05_pmc.json")))
        except Exception:  # noqa: BLE001
            pass
        if not args.no_roofline:
            sec, launches, per_pos = measure_attention(pipe.lanes[0].gpt, rows, N, T)
            nbytes, nl = attention_bytes_per_generate(gpt.config, rows, N, T)
            assert nl == launches, (nl, launches)
            ach = nbytes / sec / 1e9
            pa = pmc.get("attn_decode_kernel")
            traffic = None if pa is None else int(pa["fetch_over_algorithmic"] * nbytes / launches + pa["write_bytes_per_launch"])
            tsrc = None if pa is None else pa.get("source")
            res["roofline"] = {"bound": "hbm", "kernel": "attn_decode_persist_kernel" if rows >= 256 else "attn_decode_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                               "traffic": traffic, "traffic_source": tsrc,
                               "avg_launch_us": round(sec / launches * 1e6, 2),
                               "algorithmic_bytes_per_launch": int(nbytes / launches),
                               "launch_us_by_position": per_pos}
            gm = measure_gemms(pipe.lanes[0].gpt)
            gsched = gm.pop("_schedule")
            assert gsched == gsched0, (gsched, gsched0)   # what this leg times is what the timed region replayed
            tested = sched_tested
            nlay = gpt.config.n_layer
            tot_us = sum(us * (1 if k == "lm_head" else nlay) for k, (us, _) in gm.items())
            tot_b = sum(b * (1 if k == "lm_head" else nlay) for k, (_, b) in gm.items())
            tot_flop = 2.0 * rows * tot_b / 2          # 2 x rows x parameters (bf16: 2 bytes per parameter)
            tflops = tot_flop / tot_us / 1e6
            frac_hbm, frac_mfma = tot_b / tot_us / 1e3 / HBM_PEAK_GBS, tflops / MFMA_BF16_PEAK_TFLOPS
            # intensity = rows FLOP per weight byte against the ridge 2.5 PF / 8 TB/s = 312: below it the weight stream bounds
            res["roofline_gemm"] = {"bound": "hbm" if rows < 312 else "mfma", "kernel": f"decode-step GEMM family (M = {rows} rows)",
                                    "achieved": round(tot_b / tot_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(max(frac_hbm, frac_mfma), 4), "frac_hbm": round(frac_hbm, 4),
                                    "frac_mfma": round(frac_mfma, 4), "achieved_TFLOPs": round(tflops, 1),
                                    "mfma_peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS, "flop_per_weight_byte": rows,
                                    "weight_bytes_per_step": int(tot_b), "us_per_step": round(tot_us, 1),
                                    "us_per_step_per_128_rows": round(tot_us * 128 / rows, 1), "schedule": gsched,
                                    "schedule_tested_end_to_end": bool(tested),
                                    "schedule_source": pipe.lanes[0].gpt._engine.tile_schedule_source(),
                                    "launches_per_step": pipe.lanes[0].gpt._engine.launches_per_step(),
                                    "per_launch": {k: {"us": round(us, 2), "weight_bytes": int(b), "GB/s": round(b / us / 1e3, 1),
                                                       "TFLOPs": round(rows * b / us / 1e6, 1),
                                                       "fetch_over_algorithmic": (pmc.get("gemm", {}).get(k) or {}).get("fetch_over_algorithmic")}
                                                   for k, (us, b) in gm.items()},
                                    "traffic_source": (pmc.get("gemm") or {}).get("source")}
            # second hot kernel family (MFMA-bound): the VQ decoder's implicit-GEMM convolutions.  GFLOP per image from SURVEY.md
            # section 8d (570.1 at 384 px, 1013 + 8.6 at 512 px) x 3 split-bf16 MFMA passes, timed live over whole decode_code()
            # calls (so GroupNorm / gather time counts against it) on the current stream.
            codes = torch.randint(0, 16384, (B, N), device=dev)
            vq.decode_code(codes, [B, 8, lat, lat])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                vq.decode_code(codes, [B, 8, lat, lat])
            e1.record()
            torch.cuda.synchronize()
            vq_ms = e0.elapsed_time(e1) / 3
            tf = 3 * C["vq_gflop"] * 1e9 * B / (vq_ms * 1e-3) / 1e12
            # context for the fraction: what the vendor library's tuned bf16 GEMM reaches on THIS chip on random operands (the
            # chip clocks to its power budget: 1.19-1.27 PF measured, 1.5-1.7 PF on all-zero operands) -- the practical dense ceiling
            n = 8192
            ga, gb = torch.randn(n, n, device=dev).bfloat16(), torch.randn(n, n, device=dev).bfloat16()
            for _ in range(3):
                gc = ga @ gb
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                gc = ga @ gb
            e1.record()
            torch.cuda.synchronize()
            lib_tf = 2 * n ** 3 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
            del ga, gb, gc
            res["roofline_vq_decode"] = {"bound": "mfma", "kernel": "conv_fused_kernel (GroupNorm/swish/split fused) + AttnBlock",
                                         "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                                         "ms_per_decode_code": round(vq_ms, 2), "flop_per_image_fp32": C["vq_gflop"] * 1e9, "mfma_passes": 3,
                                         "hipblaslt_bf16_gemm_8192_random_TFLOPs": round(lib_tf, 1),
                                         "frac_of_that_measured_ceiling": round(tf / lib_tf, 4)}
            # MFMA utilisation from COUNTERS (north_star; SURVEY 8d "Reporting"): the committed SQ pass of this round (profiles/r06_sq_pmc.json,
            # tools/run_r6_sqpmc.sh -> tools/pmc_sq_summary.py): SQ_VALU_MFMA_BUSY_CYCLES over (1024 SIMDs x GRBM_GUI_ACTIVE per XCD)
            try:
                sq = json.load(open(os.path.join(ROOT, "profiles", SQ_PMC_JSON)))
                for key, fam in (("roofline_gemm", "tile"), ("roofline_vq_decode", "conv_fused_kernel")):
                    f = sq["families"].get(fam)
                    if f and rows == 640 and args.config == 2:   # the pass ran the 640-row schedule of config 2
                        res[key]["mfma_busy"] = f["mfma_busy"]
                        res[key]["mfma_busy_by_trace_at_2.4GHz"] = f.get("mfma_busy_by_trace")
                        res[key]["mfma_busy_source"] = f"profiles/{SQ_PMC_JSON}: {sq['formula']}"
            except Exception:  # noqa: BLE001 -- no committed pass: the fields stay absent
                pass
        solo_needed = world == 1 and bpc * args.lanes > 1 and not args.no_solo
        if world == 1 and not args.no_roofline and not args.no_live_traffic and "roofline" in res:
            # live counters: free the GPU memory of this process's pipeline first when the solo leg rebuilds it anyway
            lt = live_traffic(rows)
            if lt and len(lt.get("FETCH_SIZE", [])) >= 3 and len(lt.get("WRITE_SIZE", [])) >= 3:
                # tools/pmc_attn_target.py: 3 positions x L_PMC launches in dispatch order
                nlay_p = len(lt["FETCH_SIZE"]) // 3
                ratios, per_key = [], 2 * gpt.config.n_head * (gpt.config.dim // gpt.config.n_head) * 2
                for i, pos in enumerate(PMC_POSITIONS(N, T)):
                    avg = sum(lt["FETCH_SIZE"][i * nlay_p:(i + 1) * nlay_p]) / nlay_p * 2048.0
                    ratios.append(avg / ((pos + 1) * per_key * rows))
                wts = [p + 1 for p in PMC_POSITIONS(N, T)]
                ratio = sum(r * w for r, w in zip(ratios, wts)) / sum(wts)
                wr = sum(lt["WRITE_SIZE"]) / len(lt["WRITE_SIZE"]) * 1024.0
                res["roofline"]["traffic"] = int(ratio * res["roofline"]["algorithmic_bytes_per_launch"] + wr)
                res["roofline"]["traffic_source"] = ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE, then (separate pass) --pmc WRITE_SIZE, on "
                                                     "tools/pmc_attn_target.py in this bench run (FETCH x 2 gfx950 correction)")
                res["roofline"]["fetch_over_algorithmic"] = round(ratio, 4)
        solo = None
        if solo_needed:
            # ... and with ONE generate() + decode_code() of one batch in flight at a time (no cross-batch sharing at all): a fresh
            # single-lane pipeline, set up and timed after everything above
            pipe = None
            gpt._engine = None
            torch.cuda.empty_cache()
            solo = SamplingPipeline(gpt, vq, lanes=1, batches_per_chain=1)
            solo.prepare(B, N, **skw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            solo.run([make_cond() for _ in range(2)], N, **skw)
            torch.cuda.synchronize()
            res["images_per_s_with_one_step_in_flight"] = round(B * 2 / (time.perf_counter() - t1), 3)
        if world == 1 and args.config == 2 and not args.no_other_configs and not args.no_roofline:
            # BASELINE configs 4 and 5 (GPT-3B 384 px batch 64; GPT-XL t2i 512 px batch 16) in the SAME driver run: one short pass
            # each in a fresh process (own models, own HBM), after this process has released its decode state
            import gc
            pipe = solo = None
            gpt._engine = None
            gc.collect()
            torch.cuda.empty_cache()
            free_b, total_b = torch.cuda.mem_get_info()   # the children plan up to 211 GB of KV slabs + noise (config 4): what this process still holds counts
            res["other_configs"] = other_configs()
            res["other_configs"]["hbm_free_GB_when_started"] = round(free_b / 1e9, 1)
            res["other_configs"]["hbm_total_GB"] = round(total_b / 1e9, 1)
        if not args.no_cpu_baseline and world == 1:  # reported at N=1 only (the other ranks would idle in the barrier)
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


HBM_BUDGET_BYTES = 215e9   # KV slabs + noise of the chains in flight per GPU (288 GB HBM3E minus weights <= 6.2 GB, decoder activations <= 10 GB at 64 x 384 px, workspaces, slack); round 6: 180 -> 215 for two 512-row chains of GPT-XXL (196 GB) / GPT-3B (211 GB)
MFMA_BF16_PEAK_TFLOPS = 2500.0
PMC_JSON = "r06_pmc.json"
SQ_PMC_JSON = "r06_sq_pmc.json"
CPU_CAL_JSON = "r06_cpu_ref_vs_port.json"   # tools/cpu_calibrate.py, re-run every round in the build container


def PMC_POSITIONS(N, T=1):
    """cache positions of tools/pmc_attn_target.py: early / middle / last"""
    return (T - 1 + N // 9, T - 1 + N // 2, T - 1 + N - 1)


if __name__ == "__main__":
    main()
