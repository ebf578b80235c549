"""Per-instantiation register report of one translation unit (csrc/<name>.usage): VGPRs, scratch bytes, spilled VGPRs.
    python tools/usage_table.py gemm_normpre [regex on the demangled name]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "llamagen_amd", "csrc", sys.argv[1] + ".usage")).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
rows = []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    rows.append((b.split(" ")[0].strip(), g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g(r"Occupancy \[waves/SIMD\]")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
print(len(rows), "kernels;", sum(1 for r in rows if r[2] > 0), "with scratch")
for r, d in sorted(zip(rows, names), key=lambda t: t[1]):
    d = d.replace("void ", "").replace("(GemmArgs)", "")
    if r[2] > 0 or (pat and pat.search(d)):
        print(f"{d[:70]:70s} vgpr {r[1]:3d} scratch {r[2]:3d} spill {r[3]} occ {r[4]}")
