"""A few launches of ONE fused-conv shape, every weight-staging variant of lgen_conv_fused timed and compared bit for bit with variant 1
(profiling / A-B target):   python tools/conv_once.py [B] [H] [Cin] [Cout] [reps] [abl]
variants (lgen_debug_set_conv_fused_variant): 0 staging registers, 1 LDS-DMA (default), 2 / 3 LDS-DMA + pipelined fragment reads with the hi /
both pixel planes requested one tap ahead (round 6).  abl: 1 no GroupNorm / swish, 2 no residual, 4 no statistics partials, 8 zero operands."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import _lib as L
from llamagen_amd.vq_engine import _ConvW
a = sys.argv[1:]
B, H, Cin, Cout, reps, abl = (int(a[i]) if len(a) > i else d for i, d in enumerate((16, 384, 128, 128, 5, 0)))
dev = torch.device("cuda:0")
lib = L.lib()
torch.manual_seed(0)
class Cv: pass
cv = Cv(); cv.weight = (torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).to(dev); cv.bias = torch.randn(Cout, device=dev) * 0.1
cw = _ConvW(cv)
x = torch.randn(B, H, H, Cin, device=dev)
if abl & 8:  # zero operands: the DVFS / power check of the MI355X guide
    x.zero_(); cw.frag.zero_()
res = torch.randn(B, H, H, Cout, device=dev)
coef = torch.stack([1 + 0.1 * torch.randn(B, Cin, device=dev), 0.1 * torch.randn(B, Cin, device=dev)], -1).contiguous()
part = torch.empty(B, (H // 8) * (H // 16), cw.fnpad // 4, 2, device=dev)
def run(out):
    L.check(lib.lgen_conv_fused(L.ptr(x), 0 if abl & 1 else L.ptr(coef), 0 if abl & 1 else 1, L.ptr(cw.frag), L.ptr(cw.bias), 0 if abl & 2 else L.ptr(res), L.ptr(out), 0 if abl & 4 else L.ptr(part), B, H, H, Cin, Cout,
                                cw.fnpad, 3, 0, 0, L.stream()), "conv")
fl = 3 * 2 * 9 * Cin * Cout * H * H * B
ref = None
for variant in (1, 0, 2, 3, 1, 2, 3):
    lib.lgen_debug_set_conv_fused_variant(variant)
    out = torch.full((B, H, H, Cout), float("nan"), device=dev)
    part.fill_(float("nan"))
    run(out); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): run(out)
    torch.cuda.synchronize(); dt = (time.time() - t) / reps
    if ref is None:
        ref = (out.clone(), part.clone())
    same = torch.equal(out, ref[0]) and (abl & 4 or torch.equal(part, ref[1]))
    print(f"conv_fused variant {variant} abl={abl} B={B} {H}x{H} {Cin}->{Cout}: {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TFLOP/s (3-pass)  "
          f"{'bit-identical to variant 1' if same else 'DIFFERS from variant 1: max ' + format((out - ref[0]).abs().max().item(), '.3e')}", flush=True)
lib.lgen_debug_set_conv_fused_variant(1)
