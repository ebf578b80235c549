"""Cut ONE kernel out of a hipcc -save-temps device .s file, optionally patch its instruction text, and assemble it into a
stand-alone code object (.hsaco) that tools/isa_run_qkv.py loads with hipModuleLoad -- ISA-level A/B experiments with everything
else in the kernel byte-identical (round 6: the RoPE write-after-write hazard behind GPUTEST_r05).

    python tools/isa_extract.py file.s <mangled kernel name> out.hsaco [--sub 'regex' 'replacement']... [--keep-s]

Each --sub is applied (re.sub, MULTILINE) to the kernel's text only; '\\n' in the replacement inserts instructions.
Runs in the build container (clang cross-assembles gfx950 without a GPU).
"""
import os
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"


def extract(s_path, name):
    s = open(s_path).read()
    # kernel text: from its .section .text.<name> line to the next function's .section .text.* line
    m = re.search(r"^\t\.section\t\.text\." + re.escape(name) + r",.*?$", s, re.M)
    assert m, "kernel section not found"
    start = m.start()
    nxt = re.compile(r"^\t\.section\t\.text\.(?!" + re.escape(name) + r")", re.M).search(s, m.end())
    body = s[start:nxt.start() if nxt else s.index("\t.amdgpu_metadata")]
    # metadata entry of this kernel
    md0 = s.index("\t.amdgpu_metadata")
    md = s[md0:]
    entries = re.split(r"(?m)^(?=  - \.agpr_count:)", md)
    head, tail_entries = entries[0], entries[1:]
    mine = [e for e in tail_entries if re.search(r"^\s+\.name:\s+" + re.escape(name) + r"\s*$", e, re.M)]
    assert len(mine) == 1, len(mine)
    last = tail_entries[-1]
    trailer = last[last.index("amdhsa.target:"):]
    entry = mine[0]
    if "amdhsa.target:" in entry:
        entry = entry[:entry.index("amdhsa.target:")]
    header = "\t.amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"\n\t.amdhsa_code_object_version 6\n"
    return header, body, head + entry + trailer


def main():
    a = sys.argv[1:]
    s_path, name, out = a[0], a[1], a[2]
    subs = []
    i = 3
    keep = False
    while i < len(a):
        if a[i] == "--sub":
            subs.append((a[i + 1], a[i + 2].replace("\\n", "\n")))
            i += 3
        elif a[i] == "--keep-s":
            keep = True
            i += 1
        else:
            raise SystemExit("unknown argument " + a[i])
    header, body, md = extract(s_path, name)
    for pat, rep in subs:
        body, n = re.subn(pat, rep, body, flags=re.M)
        print(f"--sub {pat!r}: {n} replacement(s)")
        assert n >= 1, "pattern did not match"
    tmp_s = out + ".s"
    open(tmp_s, "w").write(header + body + md)
    obj = out + ".o"
    subprocess.check_call([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tmp_s, "-o", obj])
    subprocess.check_call([f"{LLVM}/ld.lld", "-shared", obj, "-o", out])
    os.remove(obj)
    if not keep:
        os.remove(tmp_s)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
