"""Generate golden vectors FROM THE REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py           # needs /root/reference, CPU only

Imports the reference modules from /root/reference (autoregressive/models/gpt.py,
generate.py, tokenizer/tokenizer_image/vq_model.py), loads seeded synthetic weights
(llamagen_amd.testing.synth_state_dict -- values depend only on key/shape/seed), runs the
reference's own generate() / decode_code() / VectorQuantizer.forward() on CPU and stores
small input/output vectors as .npz next to this file.  The reference has no tests or
fixtures of its own, so these files are what pins the oracle
(tests/test_oracle_golden.py) and, through injected noise, the HIP path (tests/test_gpu_*.py).

Nothing here is imported at test time; /root/reference does not exist on the GPU box.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from autoregressive.models import generate as ref_gen  # noqa: E402
from autoregressive.models.gpt import ModelArgs as RefArgs, Transformer as RefTransformer, GPT_models as RefGPT  # noqa: E402
from tokenizer.tokenizer_image.vq_model import VQ_models as RefVQ  # noqa: E402

from llamagen_amd.testing import synth_for_module  # noqa: E402
from tests.cases import FULL_DEPTH_CASES, GPT_CASES, VQ_CASES, cache_fill, full_depth_inputs, make_gpt_inputs, make_vq_inputs  # noqa: E402

torch.set_grad_enabled(False)


def build_ref_gpt(case):
    if "registry" in case:
        m = RefGPT[case["registry"]](**case["kwargs"])
    else:
        m = RefTransformer(RefArgs(**case["kwargs"]))
    sd = synth_for_module(m, seed=case["wseed"], lin_std=case.get("lin_std", 0.02))
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[case["dtype"]]
    return m.to(dtype=dt).eval()


def run_gpt_case(name, case):
    m = build_ref_gpt(case)
    cond, emb_masks = make_gpt_inputs(case)
    if case["kwargs"].get("model_type", "c2i") == "t2i":
        cond = cond.to(next(m.parameters()).dtype)
    trace = []
    orig_sample = ref_gen.sample

    def rec_sample(logits, **kw):
        trace.append(logits[:, -1, :].float().clone())
        return orig_sample(logits, **kw)

    ref_gen.sample = rec_sample
    try:
        torch.manual_seed(case["rseed"])
        out = ref_gen.generate(m, cond, case["n_new"], emb_masks=emb_masks, cfg_scale=case["cfg_scale"],
                               cfg_interval=case["cfg_interval"], temperature=case["temperature"],
                               top_k=case["top_k"], top_p=case["top_p"], sample_logits=case["sample_logits"])
    finally:
        ref_gen.sample = orig_sample
    steps = case.get("trace_steps", [0, 1, case["n_new"] - 1])
    arrs = {"tokens": out.numpy().astype(np.int32), "trace_steps": np.array(steps, dtype=np.int32),
            "trace_logits": np.stack([trace[s].numpy() for s in steps]).astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, f"gpt_{name}.npz"), **arrs)
    print(f"gpt_{name}: tokens {out.shape} first row {out[0, :8].tolist()}")


def run_full_depth_case(name, case):
    """BASELINE configs[2..4] at full depth on the REFERENCE Transformer (bf16, CPU): teacher-forced CFG-mixed logits of the
    prefill, the early positions and the late positions (K/V caches injected through the reference's own KVCache buffers)."""
    kw = case["kwargs"]
    m = build_ref_gpt(dict(case, dtype="bf16"))
    assert len(m.layers) == case["layers"]
    dt = torch.bfloat16
    cond, emb_masks, steps = full_depth_inputs(case)
    B, T = case["batch"], kw["cls_token_num"]
    N = kw["block_size"]
    if kw["model_type"] == "c2i":
        cond_c = torch.cat([cond, torch.ones_like(cond) * m.num_classes])                       # generate.py:129-131
    else:
        cond_c = torch.cat([cond.to(dt), torch.zeros_like(cond).to(dt) + m.cls_embedding.uncond_embedding])   # generate.py:136-138
    with torch.device("cpu"):
        m.setup_caches(max_batch_size=2 * B, max_seq_length=T + N, dtype=dt)                      # generate.py:147-152
    if emb_masks is not None:                                                                    # generate.py:154-163
        em = torch.cat([emb_masks, emb_masks])
        m.causal_mask[:, :, :T] = m.causal_mask[:, :, :T] * em.unsqueeze(1).to(m.causal_mask.dtype)
        eye = torch.eye(m.causal_mask.size(1), m.causal_mask.size(2))
        m.causal_mask[:] = m.causal_mask * (1 - eye) + eye
    out = {}
    H, hd = m.config.n_head, m.config.dim // m.config.n_head
    for label, tok, ipos in steps:
        if label.startswith("late"):
            p = int(ipos[0])
            S = m.layers[0].attention.kv_cache.k_cache.shape[2]
            for li, layer in enumerate(m.layers):
                kf, vf = cache_fill(2 * B, H, S, hd, p, li, dt, seed=1000 + p)
                layer.attention.kv_cache.k_cache[:, :, :p] = kf
                layer.attention.kv_cache.v_cache[:, :, :p] = vf
        if tok is None:
            logits, _ = m(None, cond_c, ipos)
        else:
            logits, _ = m(torch.cat([tok, tok]), None, ipos)
        lg = logits[:, -1].float()
        c, u = torch.split(lg, B, dim=0)
        out[label] = (u + (c - u) * case["cfg_scale"]).numpy().astype(np.float32)               # generate.py:79-84, 94-99
        print(f"full_depth_{name}: {label} max |logit| {np.abs(out[label]).max():.3f}", flush=True)
    np.savez_compressed(os.path.join(HERE, f"fulldepth_{name}.npz"), **out)


def run_vq_case(name, case):
    vq = RefVQ[case["vq"]](codebook_size=case["codebook_size"], codebook_embed_dim=case["embed_dim"]).eval()
    sd = synth_for_module(vq, seed=case["wseed"])
    vq.load_state_dict(sd)
    inp = make_vq_inputs(case)
    arrs = {}
    if case["kind"] == "decode":
        img = vq.decode_code(inp["codes"], inp["shape"])
        if case.get("store") == "sampled":   # big images: uint8 in full + every 8th pixel in fp32
            arrs["image_s8"] = img[:, :, ::8, ::8].numpy().astype(np.float32)
        else:
            arrs["image"] = img.numpy().astype(np.float32)
        arrs["uint8"] = torch.clamp(127.5 * img + 128.0, 0, 255).permute(0, 2, 3, 1).to(torch.uint8).numpy()
        print(f"vq_{name}: image {tuple(img.shape)} range [{img.min():.3f}, {img.max():.3f}]")
    elif case["kind"] == "argmin":
        _, _, (_, _, idx) = vq.quantize(inp["z"])
        arrs["indices"] = idx.numpy().astype(np.int64)
        zq = vq.quantize.get_codebook_entry(idx, list(inp["z"].shape))
        arrs["zq_head"] = zq.numpy().astype(np.float32)[:1]
        print(f"vq_{name}: indices {idx.shape} head {idx[:6].tolist()}")
    elif case["kind"] == "encode":
        with torch.no_grad():
            z = vq.quant_conv(vq.encoder(inp["x"]))
            quant, _, (_, _, idx) = vq.encode(inp["x"])
        arrs["latent"] = z.numpy().astype(np.float32)
        arrs["indices"] = idx.numpy().astype(np.int64)
        arrs["quant"] = quant.numpy().astype(np.float32)
        print(f"vq_{name}: latent {tuple(z.shape)} indices head {idx[:6].tolist()}")
    np.savez_compressed(os.path.join(HERE, f"vq_{name}.npz"), **arrs)


def run_topp_ties():
    """The reference's own top_k_top_p_filtering on rows whose nucleus boundary cuts through a tie group: how MANY entries
    survive is defined by the reference (which ones is up to its sort implementation)."""
    from tests.cases import make_topp_tie_rows
    l, tops = make_topp_tie_rows()
    kept = []
    for r in range(l.shape[0]):
        out = ref_gen.top_k_top_p_filtering(l[r:r + 1].clone(), top_k=0, top_p=tops[r])
        kept.append(int(torch.isfinite(out).sum()))
    np.savez_compressed(os.path.join(HERE, "topp_ties.npz"), kept=np.array(kept, dtype=np.int64), top_p=np.array(tops, dtype=np.float64))
    print("topp_ties: kept counts", kept)


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only or "topp_ties" in only:
        run_topp_ties()
    for name, case in GPT_CASES.items():
        if only and name not in only:
            continue
        run_gpt_case(name, case)
    for name, case in VQ_CASES.items():
        if only and name not in only:
            continue
        run_vq_case(name, case)
    for name, case in FULL_DEPTH_CASES.items():
        if only and name not in only:
            continue
        run_full_depth_case(name, case)
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump({"torch": torch.__version__, "reference": "FoundationVision/LlamaGen @ 2024_08_07",
                   "gpt_cases": list(GPT_CASES), "vq_cases": list(VQ_CASES), "full_depth_cases": list(FULL_DEPTH_CASES)}, f, indent=1)
