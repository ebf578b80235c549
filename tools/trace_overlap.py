"""Overlap analysis of a rocprofv3 --kernel-trace CSV of the bench (development aid, round 4).

Classifies every dispatch (attention / decode GEMMs / VQ decoder / sampler+embed / RNG / other), then sweeps the timeline:
  * wall time covered by each combination of classes in flight (e.g. "attn only", "attn+gemm", "gemm only", idle),
  * per class: number of dispatches, sum of durations, union of intervals (time during which >= 1 such kernel runs),
    average duration.
    python tools/trace_overlap.py <kernel_trace.csv> [t0_frac t1_frac]     (optional window of the trace, default 0.25 0.9)
"""
import csv
import json
import sys
from collections import defaultdict


def klass(name):
    n = name.lower()
    if "attn_decode" in n:
        return "attn"
    if "gemm" in n:
        return "gemm"
    if any(k in n for k in ("conv", "igemm", "gn_", "lookup", "softmax_split", "split_t", "to_uint8", "codebook")):
        return "vq"
    if any(k in n for k in ("sample", "embed", "rmsnorm", "advance")):
        return "samp"
    if "distribution" in n or "exponential" in n:
        return "rng"
    return "other"


def main():
    path = sys.argv[1]
    f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.25, 0.9)
    rows = []
    with open(path) as fh:
        rd = csv.DictReader(fh)
        for r in rd:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
    w0, w1 = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
    rows = [r for r in rows if r[0] >= w0 and r[1] <= w1]
    ev = []
    stats = defaultdict(lambda: [0, 0])
    names = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        k = klass(n)
        ev.append((s, 1, k))
        ev.append((e, -1, k))
        stats[k][0] += 1
        stats[k][1] += e - s
        short = n.split("(")[0][:70]
        names[short][0] += 1
        names[short][1] += e - s
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    combo = defaultdict(int)
    union = defaultdict(int)
    depth_time = defaultdict(int)
    last = ev[0][0]
    for t, d, k in ev:
        dt = t - last
        if dt > 0:
            key = "+".join(sorted(c for c, v in active.items() if v > 0)) or "idle"
            combo[key] += dt
            for c, v in active.items():
                if v > 0:
                    union[c] += dt
            depth_time[sum(active.values())] += dt
        active[k] += d
        last = t
    wall = ev[-1][0] - ev[0][0]
    out = dict(window_ms=wall / 1e6, dispatches=len(rows),
               combos_ms={k: round(v / 1e6, 3) for k, v in sorted(combo.items(), key=lambda kv: -kv[1])},
               combos_frac={k: round(v / wall, 4) for k, v in sorted(combo.items(), key=lambda kv: -kv[1])},
               classes={k: dict(n=v[0], sum_ms=round(v[1] / 1e6, 3), union_ms=round(union[k] / 1e6, 3), avg_us=round(v[1] / v[0] / 1e3, 2))
                        for k, v in stats.items()},
               kernels_in_flight_frac={str(k): round(v / wall, 4) for k, v in sorted(depth_time.items())},
               top_kernels={k: dict(n=v[0], avg_us=round(v[1] / v[0] / 1e3, 2), sum_ms=round(v[1] / 1e6, 2))
                            for k, v in sorted(names.items(), key=lambda kv: -kv[1][1])[:24]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
