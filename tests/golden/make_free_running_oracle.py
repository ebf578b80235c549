"""Oracle side of the bf16 free-running agreement report (tests/test_gpu_headline.py::test_bf16_free_running_agreement_report):
GPT-L (24 layers, synthetic weights of tests/test_gpu_headline.GPTL_CASE) bf16, 8 images, cfg 4.0, top-k 2000, 576 tokens from a
seeded CPU Exp(1) stream -- the oracle's token stream (fp32 accumulation), the stream of its second evaluation (fp64 accumulation
in every nn.Linear) and, per step and image, the oracle's gap between its two best log(p / q) candidates (what a first divergence
is measured against).  Five minutes of CPU; the GPU test then only runs the HIP side.
    python tests/golden/make_free_running_oracle.py      -> tests/golden/bf16_free_running_oracle.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import llamagen_oracle as O  # noqa: E402
from tests.util import build_gpt_holder, oracle_cfg  # noqa: E402

CASE = dict(registry="GPT-L", kwargs=dict(vocab_size=16384, block_size=576, num_classes=1000, cls_token_num=1, model_type="c2i"),
            wseed=21, lin_std=0.02)
B, N, V, SEED = 8, 576, 16384, 17
KW = dict(cfg_scale=4.0, cfg_interval=-1, temperature=1.0, top_k=2000, top_p=1.0, sample_logits=True)


def inputs():
    g = torch.Generator().manual_seed(SEED)
    cond = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.empty(N, B, V).exponential_(1.0, generator=g)
    return cond, noise


def main():
    torch.set_num_threads(min(16, torch.get_num_threads()))
    _, sd = build_gpt_holder(CASE)
    cfgo = oracle_cfg(CASE)
    cond, noise = inputs()
    trace = []
    qs = iter(noise)
    ref = O.generate(O.GPTOracle(cfgo, sd, torch.bfloat16), cond, N, noise_fn=lambda s: next(qs), trace=trace, **KW)
    gaps = np.zeros((N, B), dtype=np.float32)
    for f in range(N):
        lp = torch.log_softmax(O.top_k_top_p_filtering(trace[f].clone(), top_k=2000), -1)
        top2 = torch.topk(lp - torch.log(noise[f]), 2, dim=-1).values
        gaps[f] = (top2[:, 0] - top2[:, 1]).numpy()
    cache = {}
    orig = O.linear

    def lin64(x, w, dt):
        k = (w.data_ptr(), tuple(w.shape))
        if k not in cache:
            cache[k] = w.double()
        return O._rnd((x.double() @ cache[k].t()).float(), dt)
    O.linear = lin64
    try:
        qs = iter(noise)
        ref64 = O.generate(O.GPTOracle(cfgo, sd, torch.bfloat16), cond, N, noise_fn=lambda s: next(qs), **KW)
    finally:
        O.linear = orig
    out = os.path.join(ROOT, "tests", "golden", "bf16_free_running_oracle.npz")
    np.savez_compressed(out, tokens=ref.numpy(), tokens_fp64_accumulation=ref64.numpy(), top2_gap=gaps, cond=cond.numpy(),
                        noise_checksum=np.float64(noise.double().sum().item()))
    print("wrote", out, "first divergence oracle vs fp64 oracle:", [(int((a != b).float().argmax()) if bool((a != b).any()) else N) for a, b in zip(ref, ref64)])


if __name__ == "__main__":
    main()
