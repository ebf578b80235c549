#!/usr/bin/env python3
"""Headline benchmark: images/sec for LlamaGen-L 384x384 class-conditional sampling
(generate 576 tokens with CFG 4.0 / top-k 2000, then VQ decode_code) -- BASELINE.json configs[1].

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 32 images per GPU (synthetic class labels,
random-init weights of the real architecture with output.weight re-randomised, bf16 GPT + fp32-class
VQ decoder), inputs resident in HBM; N > 1 shards independent images over ranks (weak scaling, no
collective during generation) and ends every step with ONE RCCL gather of the decoded batch to rank 0.
Rank 0 prints one JSON line with `roofline` (dominant kernel = decode attention, measured live with
HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GPT_NAME, IMG, BATCH, CFG, TOPK = "GPT-L", 384, 32, 4.0, 2000
LAT = IMG // 16
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured copy)


def build_models(dev, seed):
    from llamagen_amd import GPT_models, VQ_models
    torch.manual_seed(seed)
    gpt = GPT_models[GPT_NAME](vocab_size=16384, block_size=LAT * LAT, num_classes=1000, cls_token_num=1,
                               model_type="c2i")
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


def measure_attention(gpt, cond, skw):
    """One extra eager generate() with HIP events bracketing every attention launch on its stream."""
    from llamagen_amd import generate
    eng_prof = {"events": []}
    os.environ["LGEN_NO_GRAPH"] = "1"
    try:
        gpt._prof = eng_prof
        generate(gpt, cond, LAT * LAT, **skw)
        torch.cuda.synchronize()
    finally:
        gpt._prof = None
        os.environ.pop("LGEN_NO_GRAPH", None)
    tot_ms = sum(a.elapsed_time(b) for a, b in eng_prof["events"])
    return tot_ms * 1e-3, len(eng_prof["events"])


def cpu_baseline(steps=8):
    """The CPU oracle (oracle/llamagen_oracle.py, a port of the reference path) on a bounded sample of
    the same workload: GPT-L 384 px, ONE image (CFG batch 2), prefill + `steps` decode steps at the
    END of the sequence (kv_len ~ 576, the expensive end) and `steps` at the start, plus a full
    VQ decode of one 384 px image; extrapolated to images/s over 576 tokens."""
    from llamagen_amd import GPT_models, VQ_models
    from oracle import llamagen_oracle as O
    torch.set_num_threads(os.cpu_count() or 1)
    torch.manual_seed(0)
    m = GPT_models[GPT_NAME](vocab_size=16384, block_size=LAT * LAT, num_classes=1000, cls_token_num=1, model_type="c2i")
    torch.nn.init.normal_(m.output.weight, 0, 0.02)
    sd = {k: v for k, v in m.state_dict().items()}
    cfg = O.GPTConfig(**O.GPT_SIZES[GPT_NAME], vocab_size=16384, block_size=LAT * LAT, num_classes=1000, cls_token_num=1)
    model = O.GPTOracle(cfg, sd, torch.bfloat16)
    N = LAT * LAT
    model.setup_caches(2, 1 + N)
    for kc, vc in zip(model.k_cache, model.v_cache):  # plausible cache contents for the late steps
        kc.normal_(0, 1)
        vc.normal_(0, 1)
    tok = torch.randint(0, 16384, (2, 1))
    model.forward(tok, None, torch.tensor([1]))  # untimed warm-up (first-touch of 1.4 GB of weights)
    t0 = time.time()
    for i in range(steps):  # early steps
        lg = model.forward(tok, None, torch.tensor([1 + i]))
        O.sample(O.cfg_mix(lg, CFG)[:, -1], top_k=TOPK)
    t_early = (time.time() - t0) / steps
    t0 = time.time()
    for i in range(steps):  # late steps
        lg = model.forward(tok, None, torch.tensor([N - steps + i]))
        O.sample(O.cfg_mix(lg, CFG)[:, -1], top_k=TOPK)
    t_late = (time.time() - t0) / steps
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    vsd = {k: v for k, v in vq.state_dict().items()}
    codes = torch.randint(0, 16384, (1, N))
    t0 = time.time()
    O.vq_decode_code(vsd, codes, [1, 8, LAT, LAT])
    t_vq = time.time() - t0
    per_image = 0.5 * (t_early + t_late) * N + t_vq
    return {"value": round(1.0 / per_image, 5), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle on GPT-L 384px, 1 image (CFG batch 2): {steps} early + {steps} late decode steps "
                      f"({t_early*1e3:.0f}/{t_late*1e3:.0f} ms/step) extrapolated linearly to 576 tokens + one full "
                      f"VQ decode ({t_vq:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from llamagen_amd import generate
    from llamagen_amd import dist as ldist
    import torch.distributed as dist

    rank, local, world = ldist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    seed = ldist.rank_seed(0, rank, world)
    gpt, vq = build_models(dev, seed)
    skw = dict(cfg_scale=CFG, cfg_interval=-1, temperature=1.0, top_k=TOPK, top_p=1.0, sample_logits=True)

    def step():
        c = torch.randint(0, 1000, (BATCH,), device=dev)
        idx = generate(gpt, c, LAT * LAT, **skw)
        img = vq.decode_code(idx, [BATCH, 8, LAT, LAT])
        return ldist.gather_to_root(img)  # ONE collective per step (no-op at world 1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        assert out is not None and out.shape[0] == BATCH * world and torch.isfinite(out).all()
        value = BATCH * world * args.steps / dt
        res = {"metric": "images/sec (whole node), LlamaGen-L 384px c2i", "value": round(value, 3), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "LlamaGen-L (GPT-L 343M) 384px c2i: generate 576 tokens (cfg 4.0, top-k 2000, "
                                      "bf16) + VQ-16 decode_code (fp32-class), batch 32 per GPU, random-init weights",
                          "global_batch": BATCH * world, "tokens_per_image": LAT * LAT, "parallelism": f"dp{world}"}}
        if not args.no_roofline:
            c = torch.randint(0, 1000, (BATCH,), device=dev)
            sec, launches = measure_attention(gpt, c, skw)
            nbytes, nl = attention_bytes_per_generate(gpt.config, 2 * BATCH, LAT * LAT)
            assert nl == launches, (nl, launches)
            ach = nbytes / sec / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": "attn_decode_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                               "avg_launch_us": round(sec / launches * 1e6, 2),
                               "algorithmic_bytes_per_launch": int(nbytes / launches)}
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
