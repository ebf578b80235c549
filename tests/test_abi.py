"""CPU-only: the C-ABI library loads and exports every symbol include/lgen.h declares; host logic
(packing, registries, state_dict keys, tile heuristics) without any compute call."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from llamagen_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _header_symbols():
    out = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        out += re.findall(r"^int\s+(lgen_\w+)\s*\(", txt, flags=re.M)
    return out


def test_every_declared_symbol_is_exported_and_bound(lib):
    from llamagen_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert lib.lgen_abi_version() == _lib.ABI_VERSION == 10


def test_header_arg_counts_match_ctypes_signatures():
    from llamagen_amd import _lib
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", fn)).read()
        for name, args in re.findall(r"^int\s+(lgen_\w+)\s*\(([^;]*?)\);", txt, flags=re.M | re.S):
            n = 0 if args.strip() == "void" else len([a for a in args.split(",") if a.strip()])
            assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))


def test_pack_roundtrip_and_layout():
    from llamagen_amd.engine import pack_act, pack_weight, unpack_act
    for dt, epl in ((torch.bfloat16, 8), (torch.float32, 4)):
        kc = 4 * epl
        x = torch.randn(37, 4 * kc).to(dt)
        xp = pack_act(x, 4)
        assert xp.shape == (4, 4, 4, 16, epl)
        assert torch.equal(unpack_act(xp, 37), x)
        # element (m, k) lives at [k/KC][m/16][(k%KC)/EPL][m%16][k%EPL]
        m, k = 21, kc + 2 * epl + 1
        assert xp[k // kc, m // 16, (k % kc) // epl, m % 16, k % epl] == x[m, k]
        w = torch.randn(48, 2 * kc).to(dt)
        wp = pack_weight(w)
        n = 35
        assert wp[n // 16, k // kc, (k % kc) // epl, n % 16, k % epl] == w[n, k]


def test_registries_and_missing_gpu_is_loud():
    from llamagen_amd import GPT_models, VQ_models, generate
    assert set(GPT_models) == {"GPT-B", "GPT-L", "GPT-XL", "GPT-XXL", "GPT-XXXL", "GPT-1B", "GPT-3B", "GPT-7B"}
    assert set(VQ_models) == {"VQ-16", "VQ-8"}
    from llamagen_amd.gpt import ModelArgs, Transformer
    m = Transformer(ModelArgs(n_layer=1, n_head=2, dim=64, vocab_size=64, block_size=4, num_classes=3))
    assert (m.output.weight == 0).all()  # gpt.py:305
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):  # no silent CPU fallback
            generate(m, torch.tensor([0]), 4)
        vq = VQ_models["VQ-16"]()
        with pytest.raises(RuntimeError):
            vq.decode_code(torch.zeros(1, 4, dtype=torch.long), [1, 8, 2, 2])
        from llamagen_amd.postprocess import to_uint8_hwc
        with pytest.raises(RuntimeError):
            to_uint8_hwc(torch.zeros(1, 3, 4, 4))


def test_save_npz_layout(tmp_path):
    """create_npz_from_sample_folder's output format (sample_c2i_ddp.py:21-35): one uint8 [N, H, W, 3] array `arr_0`."""
    import numpy as np
    from llamagen_amd.postprocess import save_npz
    arr = (np.arange(2 * 4 * 4 * 3) % 256).astype(np.uint8).reshape(2, 4, 4, 3)
    p = save_npz(torch.from_numpy(arr), str(tmp_path / "s.npz"))
    z = np.load(p)
    assert list(z.keys()) == ["arr_0"] and np.array_equal(z["arr_0"], arr)


def test_save_codes_format(tmp_path):
    """extract_codes_c2i.py:105-111 file layout, as dataset/imagenet.py:34-50 reads it back."""
    import numpy as np
    from llamagen_amd.postprocess import save_codes
    codes = torch.arange(2 * 9).reshape(1, 2, 9)
    cp, lp = save_codes(codes, torch.tensor([7]), str(tmp_path), "imagenet", 48, 5)
    assert cp.endswith("imagenet48_codes/5.npy") and lp.endswith("imagenet48_labels/5.npy")
    f, y = np.load(cp), np.load(lp)
    assert f.dtype == np.int64 and f.shape == (1, 2, 9) and np.array_equal(f[:, 1], codes.numpy()[:, 1]) and y.tolist() == [7]


def test_tile_heuristics_are_valid_for_every_registry_model(lib):
    """engine._tiles must hand the library a tile shape it accepts for every GPT registry size and batch
    (argument validation runs before the launch; without a GPU the launch itself then fails with a hip error,
    which is fine here -- LGEN_ERR_BAD_ARG / UNSUPPORTED is not)."""
    import types
    from llamagen_amd import _lib as L
    from llamagen_amd.engine import DecodeEngine, _ceil_div
    from llamagen_amd.gpt import find_multiple
    from oracle.llamagen_oracle import GPT_SIZES
    bad = []
    for name, sz in GPT_SIZES.items():
        d, H = sz["dim"], sz["n_head"]
        F = find_multiple(int(2 * (4 * d) / 3), 256)
        for dtype, kc in ((L.BF16, 32), (L.F32, 16)):
            for B2 in (1, 2, 6, 32, 64, 128, 256):
                mts = _ceil_div(B2, 16)
                mts = _ceil_div(mts, 8) * 8 if mts > 8 else (8 if mts > 4 else (4 if mts == 3 else mts))
                kch = d // kc
                for fuse in (False, True):
                    fuse = fuse and dtype == L.BF16 and kch % 8 == 0 and 3 <= kch // 8 <= 6
                    eng = types.SimpleNamespace(tile_override={}, pass_override={}, fuse_norm=fuse, mt=min(mts, 4), MTs=mts, kc=kc,
                                                lib=lib, dtype=torch.bfloat16 if dtype == L.BF16 else torch.float32)
                    for kind, N, K, epi in (("qkv", 3 * d, d, None), ("wo", d, d, L.EPI_RES), ("w13", 2 * F, d, L.EPI_SWIGLU),
                                            ("w2", d, F, L.EPI_RES), ("head", 16384, d, L.EPI_ROWS)):
                        mt, nt, kw = DecodeEngine._tiles(eng, kind, N, K)
                        nw = 8 if (fuse and kind in ("qkv", "w13", "head")) else 0
                        passes, _ = DecodeEngine._passes(eng, kind, N, (mt, nt, kw))   # n-groups per workgroup (explicit argument, ABI v7)
                        assert 1 <= passes <= 64 and (passes == 1 or nw), (kind, passes)
                        if kind == "qkv":
                            rc = lib.lgen_gemm_qkv_rope(8, 8, 8, 8, 8, 8, 8, B2, mts, d, H, d // H, 64 if d // H <= 64 else 128, 584, 0,
                                                        dtype, mt, nt, kw, nw, 8 if nw else 0, kch, 1e-5, passes, 0)
                        else:
                            rc = lib.lgen_gemm(8, 8, 8, B2, mts, N, K, epi, dtype, mt, nt, kw, nw, 8 if nw else 0, kch, 1e-5, 0, passes, 0)
                        if rc in (-1, -2):
                            bad.append((name, dtype, B2, fuse, kind, (mt, nt, kw), rc))
    assert not bad, bad[:10]


def test_measured_tile_tables_of_round_3(lib):
    """The shapes the round-3 sweeps measured best (profiles/r03_wide_sweep.log) are the ones engine._tiles hands out: wide models
    (GPT-3B: not fused, >= 96 k-chunks) at 128 / 256 rows, and the 4-wave RES GEMMs of the fused-norm models at 256 rows."""
    import types
    from llamagen_amd.engine import DecodeEngine

    def tiles(d, F, mts, fuse):
        eng = types.SimpleNamespace(tile_override={}, pass_override={}, fuse_norm=fuse, mt=min(mts, 4), MTs=mts, kc=32, lib=lib,
                                    dtype=torch.bfloat16)
        return {k: DecodeEngine._tiles(eng, k, n, kk) for k, n, kk in (("qkv", 3 * d, d), ("wo", d, d), ("w13", 2 * F, d),
                                                                      ("w2", d, F), ("head", 16384, d))}
    assert tiles(3200, 8704, 16, False) == {"qkv": (4, 4, 4), "wo": (4, 4, 4), "w13": (8, 2, 4), "w2": (4, 4, 8), "head": (8, 2, 4)}
    assert tiles(3200, 8704, 8, False) == {"qkv": (4, 4, 4), "wo": (4, 2, 4), "w13": (4, 4, 4), "w2": (4, 2, 8), "head": (8, 2, 4)}
    small = tiles(3200, 8704, 4, False)                                   # 64 rows: the generic rule, untouched
    assert small["w13"][0] <= 4 and small["qkv"][2] == 8
    gl = tiles(1024, 2816, 16, True)                                      # GPT-L, 256 rows
    assert gl["w2"] == (2, 2, 4) and gl["wo"] == (4, 1, 8) and gl["qkv"] == (1, 4, 8) and gl["w13"] == (2, 4, 8)
    gx = tiles(1536, 4096, 16, True)                                      # GPT-XXL: wo has 96 n-tiles and 12 chunks per wave at kw 4
    assert gx["wo"] == (2, 2, 4) and gx["w2"] == (2, 2, 4)
    assert tiles(1024, 2816, 8, True)["w2"] != (2, 2, 4)                  # 128 rows keep the round-2 shapes


def test_pinned_tile_schedules_are_shapes_the_library_takes(lib):
    """Round 5: engine.MODEL_TILE_SCHEDULES replaces the on-device search.  Every shape of every table must be one the library
    instantiates for that GEMM at the chain widths the key serves (argument validation runs before the launch: without a GPU the
    launch then fails with a hip error, which is fine here -- LGEN_ERR_UNSUPPORTED / BAD_ARG is not), the engine must pick the
    model's own table, and tile_schedule_tested() must say "tested" exactly for the keys an end-to-end test names."""
    import types
    from llamagen_amd import _lib as L
    from llamagen_amd.engine import (DecodeEngine, MODEL_TILE_SCHEDULES, TESTED_MODEL_SCHEDULES, TILE_SCHEDULES, TILE_SCHEDULE_EXACT,
                                     tile_schedule_key)
    heads = {1024: 16, 1280: 20, 1536: 24, 3200: 32}
    bad = []
    for (d, F, V), table in MODEL_TILE_SCHEDULES.items():
        H, hd = heads[d], d // heads[d]
        hdp = 64 if hd <= 64 else 128
        for key, shapes in table.items():
            widths = [key] if (table is TILE_SCHEDULES and key in TILE_SCHEDULE_EXACT) else [key, key + 8]
            for mts in widths:
                if tile_schedule_key(mts, table) != key:
                    continue
                M = mts * 16
                for kind, s in shapes.items():
                    if kind == "qkv":
                        rc = lib.lgen_gemm_qkv_rope_tile(8, 8, 8, 8, 8, 8, 8, M, mts, d, H, hd, hdp, 584, (hd + 7) // 8 * 8, L.BF16, *s, 8, 8,
                                                         d // 16, 1e-5, 0)
                    else:
                        N, K, epi, nw = {"wo": (d, d, L.EPI_RES, 0), "w13": (2 * F, d, L.EPI_SWIGLU, 8), "w2": (d, F, L.EPI_RES, 0),
                                         "head": (V, d, L.EPI_ROWS, 8)}[kind]
                        rc = lib.lgen_gemm_tile(8, 8, 8, M, mts, N, K, epi, L.BF16, *s, nw, 8 if nw else 0, d // 16, 1e-5, 0 if nw else 8, 0)
                    if rc in (-1, -2):
                        bad.append(((d, F, V), key, mts, kind, s, rc))
    assert not bad, bad

    def eng(d, F, V, mts, **kw):
        e = types.SimpleNamespace(d=d, F=F, V=V, MTs=mts, hd=64, dtype=torch.bfloat16, fuse_norm=True, use_tile=True, tile_autotune=False,
                                  tile_shape_override={}, tile_override={}, pass_override={}, pos_rows=None, mt=4, kc=32, lib=lib,
                                  _tile_refused=set())
        e.__dict__.update(kw)
        for name in ("_tile_shape", "_tiles", "_passes", "gemm_schedule", "tile_schedule_source", "tile_schedule_tested"):
            setattr(e, name, types.MethodType(getattr(DecodeEngine, name), e))
        return e
    gl = eng(1024, 2816, 16384, 40)
    assert gl._tile_shape("w13") == TILE_SCHEDULES[40]["w13"] and gl.tile_schedule_source() == "table" and gl.tile_schedule_tested()
    assert not eng(1024, 2816, 16384, 32).tile_schedule_tested()                    # 512 rows: measured, but no end-to-end oracle test
    assert eng(1024, 2816, 16384, 24).tile_schedule_tested()                        # 384 rows: the 256-row table
    assert eng(1024, 2816, 16384, 4).tile_schedule_tested()                         # 64 rows: skinny kernels
    xxl = eng(1536, 4096, 16384, 24)
    assert xxl._tile_shape("w2") == MODEL_TILE_SCHEDULES[(1536, 4096, 16384)][16]["w2"] and xxl.tile_schedule_tested()
    b3 = eng(3200, 8704, 16384, 16)
    assert b3._tile_shape("qkv") == (8, 1, 1, 6, 2, 4, 4) and b3.tile_schedule_tested()
    assert eng(1536, 4096, 16384, 32)._tile_shape("qkv") == (4, 1, 2, 6, 2, 4, 4) and eng(1536, 4096, 16384, 32).tile_schedule_tested()   # round 6: 512 rows
    b3w = eng(3200, 8704, 16384, 32)
    assert b3w._tile_shape("w2") == (2, 2, 4, 2, 2, 4, 4) and b3w._tile_shape("qkv") == (4, 1, 2, 10, 2, 4, 4) and b3w.tile_schedule_tested()
    xl = eng(1280, 3584, 16384, 16)                                                 # GPT-XL (config 5): its own table since round 6
    assert xl._tile_shape("qkv") == (4, 1, 1, 4, 4, 4, 4) and xl.tile_schedule_source() == "table" and xl.tile_schedule_tested()
    xl384 = eng(1280, 3584, 16384, 24)                                              # 2 x 12 batches of 16: its own key
    assert xl384._tile_shape("w13") == (4, 1, 2, 8, 2, 4, 4) and xl384.tile_schedule_tested() and eng(1280, 3584, 16384, 20)._tile_shape("w13") == (4, 1, 1, 8, 2, 4, 4)
    other = eng(768, 2048, 16384, 16)                                               # GPT-B: no table of its own -> GPT-L's shapes by width
    assert other._tile_shape("wo") == TILE_SCHEDULES[16]["wo"] and not other.tile_schedule_tested()
    assert other.tile_schedule_source().startswith("table (GPT-L")
    refused = eng(768, 2048, 16384, 16, _tile_refused={"w2"})                       # a shape the library refused at first launch: reported as skinny
    assert refused._tile_shape("w2") is None and refused.gemm_schedule()["w2"]["family"] == "skinny"
    assert not eng(1024, 2816, 16384, 40, tile_shape_override={"wo": (2, 2, 1, 1, 4, 4, 4)}).tile_schedule_tested()
    assert set(TESTED_MODEL_SCHEDULES) == set(MODEL_TILE_SCHEDULES)
    # K/V rows packed tighter than the lane group (round 5): head_dim 100 takes rows of 104 elements (bf16) / 100 (fp32), not 96
    for dtype, stride, want_ok in ((L.BF16, 104, True), (L.BF16, 96, False), (L.BF16, 100, False), (L.F32, 100, True), (L.BF16, 128, True)):
        rc = lib.lgen_gemm_qkv_rope(8, 8, 8, 8, 8, 8, 8, 64, 4, 3200, 32, 100, 128, 584, stride, dtype, 4, 1, 4, 0, 0, 0, 0.0, 1, 0)
        assert (rc not in (-1, -2)) == want_ok, (dtype, stride, rc)


def test_hot_kernels_have_no_register_spills():
    """The build leaves the compiler's per-kernel resource report next to every object
    (llamagen_amd/csrc/*.usage, -Rpass-analysis=kernel-resource-usage).  Every kernel of the library must be free
    of scratch memory except a short list of instantiations no default configuration launches (kept compiled
    as tuning options); the decode-loop and VQ kernels are checked by family."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "llamagen_amd", "csrc", "*.usage"))
    if not files:
        pytest.skip("no resource reports (library not built through the Makefile)")
    kernels = {}
    for f in files:
        cur = None
        for line in open(f):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = kernels.setdefault(m.group(1), {})
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    assert len(kernels) > 100
    allowed = []  # round 3: every shape that spilled is refused by its dispatcher (LGEN_ERR_UNSUPPORTED) instead of compiled
    spilled = [n for n, r in kernels.items() if r.get("ScratchSize", 0) > 0 or r.get("VGPRs Spill", 0) > 0]
    unexpected = [n for n in spilled if not any(re.search(a, n) for a in allowed)]
    assert not unexpected, unexpected
    for family in ("attn_decode_kernel", "gemm_normpre_kernel", "sample_kernel", "embed_pack_kernel", "attn_prefill"):
        members = [n for n in kernels if family in n]
        assert members, family
        assert all(kernels[n].get("ScratchSize", 0) == 0 for n in members), family


def test_device_code_has_no_crossed_packed_fp32_instruction():
    """Round 6: on MI355X `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,0]` -- a packed-fp32 multiply whose LOW
    result reads the HIGH register of a VGPR pair, made by hipcc's SLP vectoriser out of the RoPE rotation -- intermittently returned a
    wrong low product in lanes 48-63 (the wrong q elements of GPUTEST_r05; reproduced and bisected at ISA level with
    tools/isa_run_qkv.py).  The sources keep the rotation scalar (lgen_common.h: rope_pair); this test disassembles the device code
    of the BUILT library and fails when any kernel contains a packed-fp32 instruction with crossed VGPR operand selection."""
    import shutil
    from tools import isa_lint
    if not os.path.exists(isa_lint.LLVM + "/llvm-objdump"):
        pytest.skip("no llvm-objdump (the lint runs where the library is built)")
    from llamagen_amd import _lib
    hits, n_pk = isa_lint.lint(isa_lint.disassemble(_lib.LIB_PATH))
    assert n_pk > 1000, n_pk          # the disassembly really covers the device code
    assert not hits, {k: v[:2] for k, v in list(hits.items())[:5]}
    assert isa_lint.crossed(isa_lint.PK.search("v_pk_mul_f32 v[44:45], v[138:139], v[30:31] op_sel:[0,1] op_sel_hi:[0,0]"))
    assert not isa_lint.crossed(isa_lint.PK.search("v_pk_mul_f32 v[76:77], s[30:31], v[0:1] op_sel:[1,0]"))
    del shutil


def test_graft_entry_build_passes():
    """The driver's build check (__graft_entry__.build): make is a no-op on an up-to-date tree, every symbol binds, the ABI
    version of the library is the one llamagen_amd._lib expects."""
    import __graft_entry__ as g
    g.build()
