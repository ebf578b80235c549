"""Lint the DEVICE code of a built liblgen_hip.so (or of hipcc -save-temps .s files) for instruction forms this project bans.

Rule PK-CROSS (round 6): no packed-fp32 VOP3P instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose
LOW result takes a source from the HIGH register of a VGPR pair, i.e. `op_sel` bit k set for a source k that is `v[a:b]` (an SGPR
pair or a constant with op_sel is a uniform broadcast, no register-file swizzle: allowed).  On MI355X
`v_pk_mul_f32 v[44:45], v[138:139], v[30:31] op_sel:[0,1] op_sel_hi:[0,0]` (hipcc's SLP vectoriser makes it out of the RoPE
rotation `x0*c - x1*s, x1*c + x0*s`) intermittently returned a wrong LOW product in lanes 48-63 -- the wrong result of
GPUTEST_r05; tools/isa_run_qkv.py reproduces it in ~1/3 of the launches and shows that the same instruction with natural operand
selection, or two v_mul_f32, never fails (profiles/r06_isa_ab*.log).  Broadcast forms (`op_sel_hi` only) are not flagged: the low
half reads low registers there.

    python tools/isa_lint.py [path/to/liblgen_hip.so | file.s ...]
Exit code 1 (and one line per offending kernel) when the rule is violated.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\b(v_pk_(?:mul|add|fma)_f32)\s+([^\n;]*?)\s+op_sel:\[([01,]+)\]")


def crossed(m):
    """True when a VGPR-pair SOURCE operand is read with op_sel = 1 (low result <- high register)"""
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", m.group(2))]
    srcs = ops[1:]                      # ops[0] is the destination
    sel = m.group(3).split(",")
    return any(b == "1" and k < len(srcs) and srcs[k].startswith("v[") for k, b in enumerate(sel))


def disassemble(so_path):
    """yield (kernel, instruction text) for every gfx950 code object bundled in the library"""
    tmp = tempfile.mkdtemp(prefix="lgen_lint_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(so_path, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        assert cos, "no device code objects found in " + so_path
        for co in cos:
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True,
                                 capture_output=True, text=True).stdout
            kernel = "?"
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    kernel = m.group(1)
                    continue
                yield kernel, line.strip()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def from_s(path):
    kernel = "?"
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            continue
        yield kernel, line.strip()


def lint(stream):
    hits = {}
    n_pk = 0
    for kernel, text in stream:
        if "v_pk_" not in text:
            continue
        n_pk += 1
        m = PK.search(text)
        if m and crossed(m):
            hits.setdefault(kernel, []).append(text.split("//")[0].strip())
    return hits, n_pk


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sys.argv[1:] or [os.path.join(here, "llamagen_amd", "liblgen_hip.so")]
    bad = 0
    for p in paths:
        hits, n_pk = lint(from_s(p) if p.endswith(".s") else disassemble(p))
        print(f"{p}: {n_pk} packed instructions, {sum(len(v) for v in hits.values())} PK-CROSS violations in {len(hits)} kernels")
        for k, v in sorted(hits.items()):
            print("  ", k, len(v), "e.g.", v[0])
        bad += len(hits)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
