"""A few launches of ONE fused-conv shape (profiling target): python tools/conv_once.py [B] [H] [Cin] [Cout] [reps] [variant]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llamagen_amd import _lib as L
from llamagen_amd.vq_engine import _ConvW
a = sys.argv[1:]
B, H, Cin, Cout, reps = (int(a[i]) if len(a) > i else d for i, d in enumerate((16, 384, 128, 128, 5)))
variant = int(a[5]) if len(a) > 5 else 0
abl = int(a[6]) if len(a) > 6 else 0  # 1: no GroupNorm/swish, 2: no residual, 4: no statistics partials
dev = torch.device("cuda:0")
lib = L.lib()
lib.lgen_debug_set_conv_fused_variant(1 if variant == 2 else variant)
torch.manual_seed(0)
class Cv: pass
cv = Cv(); cv.weight = (torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5).to(dev); cv.bias = torch.zeros(Cout, device=dev)
cw = _ConvW(cv, wino=(variant == 2))
x = torch.randn(B, H, H, Cin, device=dev)
if abl & 8:  # zero operands: the DVFS / power check of the MI355X guide
    x.zero_(); cw.frag.zero_()
res = torch.randn(B, H, H, Cout, device=dev)
coef = torch.stack([torch.ones(B, Cin, device=dev), torch.zeros(B, Cin, device=dev)], -1).contiguous()
out = torch.empty(B, H, H, Cout, device=dev)
part = torch.empty(B, (H // 8) * (H // 16), cw.fnpad // 4, 2, device=dev)
def run_wino():
    L.check(lib.lgen_conv_wino(L.ptr(x), 0 if abl & 1 else L.ptr(coef), 0 if abl & 1 else 1, L.ptr(cw.wino), L.ptr(cw.bias), 0 if abl & 2 else L.ptr(res), L.ptr(out),
                               0 if abl & 4 else L.ptr(part), B, H, H, Cin, Cout, 0, L.stream()), "wino")
def run():
    if variant == 2:   # Winograd F(2x2, 3x3) form (LGEN_WINO_ABLATE: kernel-side ablations)
        return run_wino()
    L.check(lib.lgen_conv_fused(L.ptr(x), 0 if abl & 1 else L.ptr(coef), 0 if abl & 1 else 1, L.ptr(cw.frag), L.ptr(cw.bias), 0 if abl & 2 else L.ptr(res), L.ptr(out), 0 if abl & 4 else L.ptr(part), B, H, H, Cin, Cout,
                                cw.fnpad, 3, 0, 0, L.stream()), "conv")
run(); torch.cuda.synchronize(); t = time.time()
for _ in range(reps): run()
torch.cuda.synchronize(); dt = (time.time() - t) / reps
if variant == 2 and int(os.environ.get("LGEN_WINO_ABLATE", "0")) & 32:
    tm = part.reshape(-1)[:64].cpu().view(8, 8)
    names = ["barrier a", "halo issue + transform", "barrier b", "sub-steps 0-2 (+3's wait)", "store_halo", "sub-steps 3-15", "epilogue", "loop top"]
    tot = tm[0].sum().item()
    print("wave 0 of workgroup 0, shader clocks per phase (share):", ", ".join(f"{n} {tm[0, k].item() / tot:.3f}" for k, n in enumerate(names)), f"total {tot:.3e}")
    print("per wave totals:", [f"{tm[w].sum().item():.3e}" for w in range(8)])
fl = 3 * 2 * 9 * Cin * Cout * H * H * B
print(f"{'conv_wino' if variant == 2 else 'conv_fused'} v{variant} wino_ablate={os.environ.get('LGEN_WINO_ABLATE', 0)} abl={abl} B={B} {H}x{H} {Cin}->{Cout}: {dt*1e3:.3f} ms  {fl/dt/1e12:.0f} TFLOP/s (3-pass)")
