"""Shared helpers for the parity tests (builds oracle/HIP models for a case)."""
import os

import numpy as np
import torch

from llamagen_amd.gpt import GPT_models, ModelArgs, Transformer
from llamagen_amd.testing import synth_for_module
from llamagen_amd.vq_model import VQ_models

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def build_gpt_holder(case):
    """Our parameter-holder module (CPU) loaded with the case's synthetic weights."""
    if "registry" in case:
        m = GPT_models[case["registry"]](**case["kwargs"])
    else:
        m = Transformer(ModelArgs(**case["kwargs"]))
    sd = synth_for_module(m, seed=case["wseed"], lin_std=case.get("lin_std", 0.02))
    m.load_state_dict(sd, strict=False)
    return m.eval(), sd


def oracle_cfg(case):
    from oracle.llamagen_oracle import GPT_SIZES, GPTConfig
    kw = dict(case["kwargs"])
    if "registry" in case:
        kw.update(GPT_SIZES[case["registry"]])
    return GPTConfig(**kw)


def build_vq_holder(case):
    m = VQ_models[case["vq"]](codebook_size=case["codebook_size"], codebook_embed_dim=case["embed_dim"])
    sd = synth_for_module(m, seed=case["wseed"])
    m.load_state_dict(sd)
    return m.eval(), sd
