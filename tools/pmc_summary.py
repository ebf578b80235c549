"""rocprofv3 counter_collection.csv (FETCH_SIZE pass, WRITE_SIZE pass of tools/pmc_target.py) -> profiles/<TAG>_pmc.json and a
readable profiles/<TAG>_pmc.csv (TAG = LGEN_PMC_TAG, default r06).  FETCH_SIZE is in KB and counts HALF of a wide coalesced read stream on gfx950
(MI355X_MICROARCH.md, HBM section): bytes = KB * 1024 * 2; WRITE_SIZE: bytes = KB * 1024."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, H, hd, F, V = 1024, 16, 64, 2816, 16384
B2 = 2 * int(os.environ.get("LGEN_PMC_B", "320"))  # rows of the decode chain (tools/pmc_target.py)
TAG = os.environ.get("LGEN_PMC_TAG", "r06")
POS = (50, 300, 575)   # tools/pmc_target.py
XROW = B2 * d * 2   # bytes of one [rows, d] bf16 panel: the activation operand of a GEMM (and the residual a RES epilogue reads)
GEMM = {  # kernel-name fragment -> (bench key, algorithmic weight bytes)
    "EPI_QKV": ("wqkv", 3 * d * d * 2), "5, 4>(GemmArgs)": ("wqkv", 3 * d * d * 2),
}


def rows(pattern):
    out = []
    for f in glob.glob(pattern, recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def classify(name, grid):
    """decode-chain kernel -> (key, algorithmic bytes read: weights + activation rows [+ residual rows]) for GPT-L."""
    if "attn_decode" in name:   # attn_decode_kernel / attn_decode_persist_kernel
        return "attn", None
    if "gemm_normpre_kernel" in name or "gemm_kernel" in name or "gemm_steady_kernel" in name or "gemm_tile_kernel" in name:
        # template args: skinny <D, MT, NT, EPI, ...>, tile <D, WM, WN, MTV, NTV, KB, STAGES, EPI, NORM, LW>; EPI 5 = QKV, 4 = SWIGLU, 0 = ROWS, 3 = RES
        args = name[name.index("<") + 1:name.index(">")].split(",")
        epi = int(args[7 if "gemm_tile_kernel" in name else 3])
        if epi == 5:
            return "wqkv", 3 * d * d * 2 + XROW
        if epi == 4:
            return "w13", 2 * F * d * 2 + XROW
        if epi == 0:
            return "lm_head", V * d * 2 + XROW
        if epi == 3:
            return "res", None  # wo (d x d) and w2 (d x F) share the instantiation: split by grid size below
    return None, None


def main(fetch_dir, write_dir):
    res = {"gemm": {"source": f"profiles/{TAG}_pmc.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_target.py)"}}
    table = []
    for counter, ddir, scale in (("FETCH_SIZE", fetch_dir, 2048.0), ("WRITE_SIZE", write_dir, 1024.0)):
        acc = {}
        attn_seq = []
        res_seq = []
        for r in rows(os.path.join(ddir, "**", "*counter_collection.csv")):
            if r["Counter_Name"] != counter:
                continue
            key, alg = classify(r["Kernel_Name"], r["Grid_Size"])
            if key is None:
                continue
            val = float(r["Counter_Value"]) * scale
            if key == "attn":
                attn_seq.append((int(r["Dispatch_Id"]), val))
                continue
            if key == "res":  # wo and w2 share instantiation and grid (N = d for both): they alternate in dispatch order
                res_seq.append((int(r["Dispatch_Id"]), val))
                continue
            a = acc.setdefault(key, [0, 0.0, alg])
            a[0] += 1
            a[1] += val
        res_seq.sort()
        for i, (_, val) in enumerate(res_seq):  # per layer: ... attention, wo (RES), w1||w3, w2 (RES) ...
            key, alg = ("wo", d * d * 2 + 2 * XROW) if i % 2 == 0 else ("w2", F * d * 2 + B2 * F * 2 + XROW)   # + operand rows + residual rows
            a = acc.setdefault(key, [0, 0.0, alg])
            a[0] += 1
            a[1] += val
        for key, (n, tot, alg) in sorted(acc.items()):
            table.append((counter, key, n, tot / n, alg, tot / n / alg))
            if counter == "FETCH_SIZE":
                res["gemm"].setdefault(key, {})["fetch_over_algorithmic"] = round(tot / n / alg, 4)
                res["gemm"][key]["fetch_bytes_per_launch"] = int(tot / n)
            else:
                res["gemm"].setdefault(key, {})["write_bytes_per_launch"] = int(tot / n)
        # attention: the LAST 72 attention dispatches are the 3 x 24 full-size launches at positions POS
        attn_seq.sort()
        last = [v for _, v in attn_seq[-72:]]
        if len(last) == 72:
            ratios = []
            for i, pos in enumerate(POS):
                avg = sum(last[i * 24:(i + 1) * 24]) / 24
                alg = (pos + 1) * 2 * H * hd * 2 * B2
                table.append((counter, f"attn pos {pos}", 24, avg, alg, avg / alg if counter == "FETCH_SIZE" else avg / (B2 * d * 2)))
                ratios.append(avg / alg)
            if counter == "FETCH_SIZE":
                # weight the three positions like a generate() does (bytes grow linearly with position)
                res["attn_decode_kernel"] = {"fetch_over_algorithmic": round(sum(r * (p + 1) for r, p in zip(ratios, POS)) / sum(p + 1 for p in POS), 4),
                                             "fetch_over_algorithmic_by_position": {str(p): round(r, 4) for p, r in zip(POS, ratios)}}
            else:
                res.setdefault("attn_decode_kernel", {})["write_bytes_per_launch"] = int(sum(last) / 72)
    res.setdefault("attn_decode_kernel", {})["source"] = res["gemm"]["source"]
    res["rows"] = B2
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", f"{TAG}_pmc.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate pass, WRITE_SIZE) -- python tools/pmc_target.py ; GPT-L bf16, B2 = {B2}\n")
        f.write("# FETCH_SIZE KB x 1024 x 2 (gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md); WRITE_SIZE KB x 1024\n")
        f.write("counter,kernel,launches,bytes_per_launch,algorithmic_bytes,ratio\n")
        for t in table:
            f.write(f"{t[0]},{t[1]},{t[2]},{t[3]:.0f},{t[4]},{t[5]:.4f}\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
