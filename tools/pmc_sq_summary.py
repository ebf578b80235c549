"""rocprofv3 counter_collection.csv of an SQ counter pass -> per-kernel averages (development aid, round 3).
    python tools/pmc_sq_summary.py <dir> [<dir> ...]  > table
Groups dispatches by (kernel name, grid size) and prints the mean of every counter plus derived shares:
WAIT_ANY / WAVE_CYCLES (waves parked on s_waitcnt / barrier), ACTIVE_INST_VALU / WAVE_CYCLES, ...  SQ_*_CYCLES count
quad-cycles per wave (MI355X_MICROARCH.md)."""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    for a, b in (("gemm_normpre_kernel", "normpre"), ("gemm_kernel", "gemm"), ("attn_decode_kernel", "attn"),
                 ("(GemmArgs)", ""), ("(AttnArgs)", "")):
        name = name.replace(a, b)
    return name[:46]


def main(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Grid_Size"])
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    names = sorted({c for v in acc.values() for c in v})
    print("kernel,grid,launches," + ",".join(names))
    for k, v in sorted(acc.items(), key=lambda kv: -max(a[0] for a in kv[1].values())):
        n = max(a[0] for a in v.values())
        if n < 20:
            continue
        print(f"\"{k[0]}\",{k[1]},{n}," + ",".join(f"{v[c][1] / max(1, v[c][0]):.0f}" if c in v else "" for c in names))
        wc = v.get("SQ_WAVE_CYCLES")
        if wc and wc[1] > 0:
            print("  shares of SQ_WAVE_CYCLES: " + "  ".join(f"{c[3:]}={v[c][1] / wc[1]:.3f}" for c in names
                                                             if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_MFMA", "SQ_INSTS_LDS")))


if __name__ == "__main__":
    main(sys.argv[1:])
