"""ctypes binding of liblgen_hip.so (C ABI: include/lgen.h).

The HIP library is the product path: if it cannot be loaded this module raises -- there is no
CPU / PyTorch fallback anywhere in llamagen_amd.  `import torch` must happen first so that the
library's libamdhip64.so.7 dependency binds to the HIP runtime torch already loaded (one runtime
per process; SURVEY.md section 7 "ROCm runtime duality").
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblgen_hip.so")
BF16, F32, F16 = 0, 1, 2
EPI_ROWS, EPI_PACKED, EPI_GELU, EPI_RES, EPI_SWIGLU, EPI_QKV = 0, 1, 2, 3, 4, 5
ABI_VERSION = 10
SSQ_STRIDE = 256  # LGEN_SSQ_STRIDE: floats per row of a fused-RMSNorm statistics array
ERR_UNSUPPORTED = -2
ERR_BAD_ARG = -1

_c = ctypes
_P, _I, _F = _c.c_void_p, _c.c_int, _c.c_float

# name -> argtypes; the single source of truth for tests/test_abi.py as well
SIGNATURES = {
    "lgen_abi_version": [],
    "lgen_embed_pack": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lgen_ssq_pack": [_P, _P, _I, _I, _I, _P],
    "lgen_ssq_group4": [_P, _P, _I, _I, _P],
    "lgen_rmsnorm": [_P, _P, _P, _I, _I, _F, _I, _P],
    "lgen_gemm": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _F, _P, _I, _P],
    "lgen_gemm_max_kw": [_I, _I, _I, _I],
    "lgen_gemm_qkv_rope": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _F, _I, _P],
    "lgen_gemm_tile": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _F, _P, _P],
    "lgen_gemm_qkv_rope_tile": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _F, _P],
    "lgen_resize_bicubic": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lgen_to_uint8_hwc": [_P, _P, _I, _I, _I, _I, _P],
    "lgen_debug_set_igemm_variant": [_I],
    "lgen_debug_set_prefill_mfma": [_I],
    "lgen_debug_set_conv_fused_variant": [_I],
    "lgen_debug_set_vq_nt": [_I],
    "lgen_attn_decode": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgen_rope_append_prefill": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgen_attn_prefill": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgen_sample": [_P, _P, _c.c_longlong, _P, _P, _P, _I, _I, _I, _I, _F, _I, _F, _I, _F, _I, _I, _P],
    "lgen_advance_state": [_P, _P],
    "lgen_embed_rows": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "lgen_gemm_qkv_rope_rows": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _F, _I, _P],
    "lgen_attn_decode_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgen_sample_rows": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _F, _I, _F, _I, _I, _P],
    "lgen_vq_codebook_prep": [_P, _P, _P, _I, _I, _I, _P],
    "lgen_vq_lookup_pqconv": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lgen_vq_argmin": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lgen_gn_stats": [_P, _P, _P, _I, _I, _I, _F, _I, _P],
    "lgen_gn_swish_split": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lgen_split_t": [_P, _P, _P, _I, _I, _I, _I, _P],
    "lgen_softmax_split": [_P, _P, _P, _I, _I, _I, _P],
    "lgen_conv_fused_bn": [_I],
    "lgen_conv_fused": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgen_gn_finalize": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "lgen_conv_igemm": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _c.c_longlong, _F, _P],
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile liblgen_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(os.cpu_count() or 4)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building liblgen_hip.so failed")
    return LIB_PATH


def register(sigs):
    SIGNATURES.update(sigs)
    if _lib is not None:
        _bind(_lib, sigs)


def _bind(lib, sigs):
    for name, args in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.argtypes = args
        fn.restype = _I


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(llamagen_amd has no CPU/PyTorch fallback path)")
        l = ctypes.CDLL(LIB_PATH)
        _bind(l, SIGNATURES)
        _lib = l
    return _lib


class LgenError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: not supported by the HIP library yet")
    if rc != 0:
        raise LgenError(f"{what} failed with code {rc} ({'bad argument' if rc == -1 else 'hipError'})")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
