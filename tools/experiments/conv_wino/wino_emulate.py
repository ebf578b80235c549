"""CPU emulation of the VQ decoder with its 3x3 convolutions in Winograd form (development aid, round 5; quoted by csrc/conv_wino.hip).

Every 3x3 convolution with >= 128 output channels of the oracle's decoder is replaced by Y = A^T [(G g G^T) (.) (B^T d B)] A with
fp32 transforms and the 3-pass split-bf16 product (hi*hi + hi*lo + lo*hi, fp32 accumulate) on the transformed operands; the other
convolutions keep the direct split-bf16 form.  Prints the max / mean absolute distance of the decoded image from the fp32 decoder.
    python tools/wino_emulate.py [latent side, default 8] [direct | f23 | f43]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import VQ_models  # noqa: E402
from llamagen_amd.testing import synth_for_module  # noqa: E402
from oracle import llamagen_oracle as O  # noqa: E402

FORMS = {
    "f23": (torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
            torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32),
            torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32), 4, 2),
    "f43": (torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                          [0, 0, 1]], dtype=torch.float64),
            torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                          [0, 4, 0, -5, 0, 1]], dtype=torch.float32),
            torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32), 6, 4),
}


def split(x):
    hi = x.to(torch.bfloat16).float()
    return hi, (x - hi).to(torch.bfloat16).float()


def main():
    lat = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mode = sys.argv[2] if len(sys.argv) > 2 else "f23"
    torch.manual_seed(0)
    vq = VQ_models["VQ-16"](codebook_size=16384, codebook_embed_dim=8)
    sd = synth_for_module(vq, seed=3)
    codes = torch.randint(0, 16384, (1, lat * lat), generator=torch.Generator().manual_seed(0))
    ref = O.vq_decode_code(sd, codes, [1, 8, lat, lat])

    def direct(x, w, b, padding):
        xh, xl = split(x)
        wh, wl = split(w)
        return F.conv2d(xh, wh, None, padding=padding) + F.conv2d(xh, wl, None, padding=padding) + F.conv2d(xl, wh, b, padding=padding)

    def conv(x, sd_, name, padding):
        w, b = sd_[name + ".weight"], sd_[name + ".bias"]
        if mode == "direct" or w.shape[-1] != 3 or w.shape[0] < 128:
            return direct(x, w, b, padding)
        G, BT, AT, n, m = FORMS[mode]
        Bn, C, H, W = x.shape
        K = w.shape[0]
        U = (G @ w.double() @ G.t()).float()
        pt = F.pad(x, (1, 1, 1, 1)).unfold(2, n, m).unfold(3, n, m)
        V = torch.einsum('ij,bchwjk,lk->bchwil', BT, pt, BT)
        Uh, Ul = split(U)
        Vh, Vl = split(V)
        prod = lambda u, v: torch.einsum('kcil,bchwil->bkhwil', u, v)
        M = prod(Uh, Vh) + prod(Uh, Vl) + prod(Ul, Vh)
        Y = torch.einsum('pi,bkhwil,ql->bkhwpq', AT, M, AT)
        return Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, K, H, W) + b.view(1, -1, 1, 1)

    O._conv = conv
    got = O.vq_decode_code(sd, codes, [1, 8, lat, lat])
    d = (got - ref).abs()
    print(f"{mode}: {16 * lat} px image, max abs err {d.max().item():.3e}, mean {d.mean().item():.3e} (reference |max| {ref.abs().max().item():.2f})")


if __name__ == "__main__":
    main()
