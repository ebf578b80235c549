"""Diagnostic (round 3): where do the few fused-norm lm_head outputs at 256 rows differ from the oracle by more than 1 ulp?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llamagen_amd import _lib as L
from llamagen_amd.engine import pack_act, pack_weight
from oracle import llamagen_oracle as O

dev = torch.device("cuda:0")
lib = L.lib()
dt, d, V = torch.bfloat16, 1024, 4096


def rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt)


for M in (128, 256):
    mts = M // 16
    x = rand((M, d), 51, 1.3)
    nw = (1 + 0.1 * rand((d,), 52).float()).to(dt)
    wh = rand((V, d), 56, 0.03)
    xp, nw_d, whp = pack_act(x.to(dev), mts), nw.to(dev), pack_weight(wh.to(dev))
    ssq = torch.full((mts * 16, L.SSQ_STRIDE), float("nan"), device=dev)
    L.check(lib.lgen_ssq_pack(L.ptr(xp), L.ptr(ssq), mts, d, L.BF16, L.stream()), "ssq_pack")
    rows = torch.zeros(mts * 16, V, dtype=dt, device=dev)
    L.check(lib.lgen_gemm(L.ptr(whp), L.ptr(xp), L.ptr(rows), M, mts, V, d, L.EPI_ROWS, L.BF16, 2, 4, 8, L.ptr(nw_d), L.ptr(ssq),
                          d // 16, 1e-5, 0, L.stream()), "head")
    torch.cuda.synchronize()
    got = rows[:M].float().cpu()
    xn = O.rms_norm(x.float(), nw, 1e-5, dt)
    ref = O.linear(xn, wh.float(), dt)
    ex = (xn.double() @ wh.double().t())                 # exact-ish pre-rounding value from the ORACLE's xn
    r64 = ex.float().to(dt).float()
    ss = ssq[:M, :d // 16].sum(1).cpu()
    ss_ref = (x.float() ** 2).sum(-1)
    print(f"M={M}: ssq rel err max {((ss - ss_ref).abs() / ss_ref).max():.2e}")
    for name, a in (("hip", got), ("oracle", ref)):
        e = (a - r64).abs()
        ulp = torch.maximum(a.abs(), r64.abs()) * 2.0 ** -7 + 1e-4
        bad = e > 1.01 * ulp
        print(f"  {name} vs fp64-of-oracle-xn: mismatching {int((e > 0).sum())}, beyond 1 ulp {int(bad.sum())}, max {e.max():.4f}")
    e = (got - ref).abs()
    bad = e > 1.01 * (torch.maximum(got.abs(), ref.abs()) * 2.0 ** -7 + 1e-4)
    idx = bad.nonzero()
    print(f"  hip vs oracle: beyond 1 ulp {len(idx)}; rows {sorted(set(idx[:, 0].tolist()))[:20]}")
    for r, c in idx[:8].tolist():
        print(f"    [{r},{c}] hip {got[r, c]:.5f} oracle {ref[r, c]:.5f} exact {ex[r, c]:.6f}")
    # the HIP kernel's own normalised activations, reconstructed: solve from an identity-weight GEMM
    eye = torch.eye(d, dtype=dt)
    eyep = pack_weight(eye.to(dev))
    xrows = torch.zeros(mts * 16, d, dtype=dt, device=dev)
    L.check(lib.lgen_gemm(L.ptr(eyep), L.ptr(xp), L.ptr(xrows), M, mts, d, d, L.EPI_ROWS, L.BF16, 2, 4, 8, L.ptr(nw_d), L.ptr(ssq),
                          d // 16, 1e-5, 0, L.stream()), "identity")
    torch.cuda.synchronize()
    xn_hip = xrows[:M].float().cpu()
    dx = (xn_hip - xn).abs()
    print(f"  normalised activations hip vs oracle: differing {int((dx > 0).sum())} of {dx.numel()}, max {dx.max():.5f}, rows with diffs "
          f"{sorted(set((dx > 0).nonzero()[:, 0].tolist()))[:16]}")
