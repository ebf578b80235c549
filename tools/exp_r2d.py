"""GEMM tile shapes for WIDE decode chains (96 / 128 images per chain = 192 / 256 rows): decode-only img/s per LGEN_TILES set."""
import os, sys
sys.argv = [sys.argv[0]]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_r2a as E
gpt = E.build()
SETS = {
    "default": "",
    "mt4": "qkv=4,2,8;wo=4,1,8;w13=4,2,8;w2=4,1,8;head=4,4,8",
    "mt8": "qkv=8,1,8;wo=8,1,8;w13=8,2,8;w2=8,1,8;head=8,2,8",
    "mt4nt4": "qkv=4,2,8;wo=4,2,8;w13=4,4,8;w2=4,2,8;head=4,4,8",
    "mt8nt2": "qkv=8,2,4;wo=8,2,4;w13=8,2,4;w2=8,2,8;head=8,2,8",
    "mt2": "qkv=2,4,8;wo=2,1,8;w13=2,4,8;w2=2,1,8;head=4,4,8",
}
for name, spec in SETS.items():
    os.environ["LGEN_TILES"] = spec
    for norm in ("1", "0"):
        os.environ["LGEN_FUSED_NORM"] = norm
        try:
            E.e2(gpt, [(96, 1)], tag=f"E4 {name} fused_norm={norm}")
        except Exception as ex:
            print(f"E4 {name} fused_norm={norm}: failed {type(ex).__name__}: {str(ex)[:80]}", flush=True)
