"""rocprofv3 counter_collection.csv (+ kernel_trace.csv) of an SQ counter pass -> per-kernel averages and MFMA utilisation.
    python tools/pmc_sq_summary.py <dir> [<dir> ...] [--json out.json]  > table.csv
Groups dispatches by (kernel name, grid size): mean of every counter, mean duration from the kernel trace, and
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)      (gfx94x MfmaUtil formula;
    rocprofv3 on ROCm 7.2 has no gfx950 derived-counter section, MI355X_MICROARCH.md "rocprofv3 PMC slots").  rocprofv3 reports
    GRBM_GUI_ACTIVE SUMMED over the 8 XCDs (17.07 "cycles per ns" on a 3.7 ms kernel = 8 x 2.13 GHz) and SQ_VALU_MFMA_BUSY_CYCLES
    summed over all SIMDs, 16 cycles per v_mfma_f32_16x16x32_bf16 (checked: 4 076 863 488 / 254 803 968 SQ_INSTS_MFMA = 16.0, and
    254 803 968 = 32 x 384^2 x 128 x 128 x 9 x 3 passes / 8192 MACs, the 128 -> 128 conv at 384 px exactly).  For kernels of a few
    us GRBM_GUI_ACTIVE also covers the dispatch around the kernel (30 "cycles per ns"), so the table carries a second figure,
    mfma_busy_by_trace = the same cycles over (1024 SIMDs x trace duration x 2.4 GHz), a LOWER bound (the chip rarely holds 2.4 GHz).
plus the shares of SQ_WAVE_CYCLES (quad-cycles per wave) spent issuing VALU / waiting.  Counter values are the sums rocprofv3
reports over all shader engines / XCDs; GRBM_GUI_ACTIVE is checked against the trace duration (cycles per ns printed per kernel)."""
import csv, glob, json, os, sys
from collections import defaultdict

N_SIMD = 4 * 256
N_XCD = 8
F_MAX_GHZ = 2.4


def short(name):
    name = name.replace("void ", "")
    for a, b in (("gemm_normpre_kernel", "normpre"), ("gemm_tile_kernel", "tile"), ("gemm_kernel", "gemm"),
                 ("attn_decode_persist_kernel", "attn_persist"), ("attn_decode_kernel", "attn"), ("(GemmArgs)", ""), ("(AttnArgs, int, int)", ""),
                 ("(AttnArgs)", ""), ("(anonymous namespace)::", ""), ("(ConvFArgs)", "")):
        name = name.replace(a, b)
    return name[:64]


def main(argv):
    out_json = None
    if "--json" in argv:
        out_json = argv[argv.index("--json") + 1]
        argv = [a for a in argv if a not in ("--json", out_json)]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for d in argv:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Grid_Size"])
                a = acc[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r.get("Grid_Size") or str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))))
                dur[k][0] += 1
                dur[k][1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    names = sorted({c for v in acc.values() for c in v})
    print("kernel,grid,launches,avg_us,mfma_busy,mfma_busy_by_trace,grbm_ghz_per_xcd," + ",".join(names))
    rec = {}
    for k, v in sorted(acc.items(), key=lambda kv: -sum(a[1] for c, a in kv[1].items() if c == "GRBM_GUI_ACTIVE")):
        n = max(a[0] for a in v.values())
        if n < 4:
            continue
        mean = {c: v[c][1] / max(1, v[c][0]) for c in names if c in v}
        us = dur[k][1] / dur[k][0] / 1e3 if dur[k][0] else float("nan")
        grbm = mean.get("GRBM_GUI_ACTIVE", 0.0)
        busy = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * grbm / N_XCD) if grbm else float("nan")
        busy_t = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (N_SIMD * us * 1e3 * F_MAX_GHZ) if us == us and us > 0 else float("nan")
        print(f"\"{k[0]}\",{k[1]},{n},{us:.2f},{busy:.4f},{busy_t:.4f},{grbm / N_XCD / (us * 1e3) if us == us and us > 0 else float('nan'):.3f},"
              + ",".join(f"{mean[c]:.0f}" if c in mean else "" for c in names))
        wc = mean.get("SQ_WAVE_CYCLES")
        if wc:
            print("  shares of SQ_WAVE_CYCLES: " + "  ".join(f"{c[3:]}={mean[c] / wc:.3f}" for c in names if c.startswith("SQ_") and c not in
                                                             ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_MFMA", "SQ_INSTS_LDS",
                                                              "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")))
        rec[f"{k[0]} grid {k[1]}"] = dict(launches=n, avg_us=round(us, 2), mfma_busy=round(busy, 4), mfma_busy_by_trace=round(busy_t, 4), **{c: round(m) for c, m in mean.items()})
    if out_json:
        fam = {}
        for name, r in rec.items():
            for f in ("conv_fused_kernel", "tile<", "attn_persist", "igemm_kernel"):
                if f in name:
                    a = fam.setdefault(f.strip("<"), dict(mfma_cycles=0.0, grbm=0.0, us=0.0))
                    a["mfma_cycles"] += r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) * r["launches"]
                    a["grbm"] += r.get("GRBM_GUI_ACTIVE", 0) * r["launches"]
                    a["us"] += r["avg_us"] * r["launches"]
        for f, a in fam.items():
            a["mfma_busy"] = round(a["mfma_cycles"] / (N_SIMD * a["grbm"] / N_XCD), 4) if a["grbm"] else None
            a["mfma_busy_by_trace"] = round(a["mfma_cycles"] / (N_SIMD * a["us"] * 1e3 * F_MAX_GHZ), 4) if a["us"] else None
            a["effective_clock_ghz"] = round(a["grbm"] / N_XCD / (a["us"] * 1e3), 3) if a["us"] else None
        json.dump(dict(source="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS "
                              "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python tools/pmc_sq_target.py",
                       formula="mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), time-weighted over the family's launches; mfma_busy_by_trace = the same cycles / (1024 SIMDs x kernel-trace duration x 2.4 GHz), a lower bound (GRBM_GUI_ACTIVE of a few-us kernel also covers its dispatch)",
                       families=fam, kernels=rec), open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
