"""Host side of the Winograd F(2x2, 3x3) convolution (llamagen_amd/vq_engine.py: _ConvW.wino; csrc/conv_wino.hip), on CPU:
the transformed-weight packing against the index formula the kernel reads with, and the arithmetic of the form (fp32 transforms,
3-pass split-bf16 products on the TRANSFORMED operands) against fp64 conv2d at the tolerance the GPU test uses."""
import numpy as np
import torch
import torch.nn.functional as F

from llamagen_amd.vq_engine import _ConvW


class _Cv:
    pass


def _conv(cout, cin, seed):
    g = torch.Generator().manual_seed(seed)
    cv = _Cv()
    cv.weight = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    cv.bias = 0.1 * torch.randn(cout, generator=g)
    return cv


def _split(x):
    hi = x.to(torch.bfloat16).float()
    return hi, (x - hi).to(torch.bfloat16).float()


def test_wino_weight_packing_matches_the_kernels_index_formula():
    """conv_wino_kernel: wave w = pi * 2 + wn requests, in sub-step ss = pj * 4 + jt of chunk kc of block nb, the 2 x 1 KiB at
    ((((nb * nkc + kc) * 16 + ss) * 8 + w) * 2 + plane) KiB and reads lane (g, r) of it as the 8 input channels kc * 32 + g * 8 ..
    of output channel nb * 128 + (wn * 4 + jt) * 16 + r of position pi * 4 + pj of U = G g G^T."""
    cout, cin = 256, 64
    cv = _conv(cout, cin, 1)
    cw = _ConvW(cv, wino=True)
    assert _ConvW(cv, wino=False).wino is None and cw.wino is not None
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    U = (G @ cv.weight.double() @ G.t()).float()              # [cout][cin][4][4]
    Uh, Ul = _split(U)
    flat = cw.wino.reshape(-1).float()                        # bf16 elements in kernel order
    nkc = cin // 32
    assert flat.numel() == 2 * 16 * cout * cin
    rng = np.random.default_rng(0)
    for _ in range(400):
        nb, kc, ss, w, plane = rng.integers(cout // 128), rng.integers(nkc), rng.integers(16), rng.integers(8), rng.integers(2)
        g, r, e = rng.integers(4), rng.integers(16), rng.integers(8)
        pj, jt, pi, wn = ss >> 2, ss & 3, w >> 1, w & 1
        off = ((((((nb * nkc + kc) * 16 + ss) * 8 + w) * 2 + plane) * 4 + g) * 16 + r) * 8 + e
        co, ci = nb * 128 + (wn * 4 + jt) * 16 + r, kc * 32 + g * 8 + e
        want = (Uh if plane == 0 else Ul)[co, ci, pi, pj]
        assert float(flat[off]) == float(want), (nb, kc, ss, w, plane, g, r, e)


def test_wino_form_numerics_match_the_direct_split_form():
    """Y = A^T [(G g G^T) (.) (B^T d B)] A with fp32 transforms and hi*hi + hi*lo + lo*hi products on the transformed operands
    (what the kernel computes, in the order of its epilogue: column half, then row half) against fp64 conv2d: < 1e-4 of the output
    scale (tests/test_gpu_vq.py::test_conv_wino_vs_fp32_reference uses the same bar)."""
    g = torch.Generator().manual_seed(3)
    B, C, K, H, W = 2, 64, 128, 16, 32
    x = torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3
    cv = _conv(K, C, 2)
    ref = F.conv2d(x.double(), cv.weight.double(), cv.bias.double(), padding=1).float()
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    U = (G @ cv.weight.double() @ G.t()).float()
    pt = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)                # [B][C][H/2][W/2][4][4]
    V = torch.einsum('ij,bchwjk,lk->bchwil', BT, pt, BT)
    Uh, Ul = _split(U)
    Vh, Vl = _split(V)
    prod = lambda u, v: torch.einsum('kcil,bchwil->bkhwil', u, v)
    M = prod(Ul, Vh) + prod(Uh, Vl) + prod(Uh, Vh)                              # the kernel's MFMA order: lo*hi, hi*lo, hi*hi
    cpart = torch.einsum('bkhwil,ql->bkhwiq', M, AT)                            # column half (over pj), per position row pi
    Y = torch.einsum('pi,bkhwiq->bkhwpq', AT, cpart)                            # row half (over pi)
    got = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, H, W) + cv.bias.view(1, -1, 1, 1)
    err = (got - ref).abs().max().item()
    assert err < 1e-4 * max(1.0, ref.abs().max().item()), err
    # ... and it is the same function as the direct 3-pass split form up to the rounding of the transformed operands
    xh, xl = _split(x)
    wh, wl = _split(cv.weight)
    direct = F.conv2d(xh, wh, None, padding=1) + F.conv2d(xh, wl, None, padding=1) + F.conv2d(xl, wh, cv.bias, padding=1)
    assert (got - direct).abs().max().item() < 1.2e-4 * max(1.0, ref.abs().max().item())
