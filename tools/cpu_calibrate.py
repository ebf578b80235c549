"""Same-host calibration of bench.py's CPU leg (round 4): the reference's own modules (/root/reference) and the oracle port, run
back to back in the build container on the same bounded slice (bench.cpu_baseline: 32-step B = 32 slice + config 1 in full) with
the same thread count.  Writes profiles/r06_cpu_ref_vs_port.json (round 6 re-run; r04_cpu_ref_vs_port.json is the round-4 pass); bench.py quotes `port_over_reference` next to a `kind: "port"`
figure on boxes without the reference mount.
    python tools/cpu_calibrate.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import json, bench; print('CPU_LEG ' + json.dumps(bench.cpu_baseline()))"


def leg(port):
    env = dict(os.environ, LGEN_BENCH_CPU_PORT="1" if port else "0")
    r = subprocess.run([sys.executable, "-c", CODE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
    line = [l for l in r.stdout.splitlines() if l.startswith("CPU_LEG ")]
    if not line:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-2000:])
    return json.loads(line[-1][8:])


def main():
    assert os.path.isdir("/root/reference"), "the calibration needs the reference mount (build container)"
    ref, port = leg(False), leg(True)
    assert ref["kind"] == "reference" and port["kind"] == "port", (ref["kind"], port["kind"])
    path = os.path.join(ROOT, "profiles", "r06_cpu_ref_vs_port.json")
    this = {"slice": round(port["value"] / ref["value"], 3), "c1": round(port["c1"]["value"] / ref["c1"]["value"], 3),
            "reference_slice_images_per_s": ref["value"], "port_slice_images_per_s": port["value"],
            "reference_c1": ref["c1"]["value"], "port_c1": port["c1"]["value"]}
    passes = []
    try:   # the build container's host is shared: every pass is kept and the MEDIAN of the ratios is what bench.py quotes
        passes = json.load(open(path)).get("passes", [])
    except Exception:  # noqa: BLE001
        pass
    passes.append(this)
    med = lambda k: sorted(p[k] for p in passes)[len(passes) // 2]
    out = {"host_logical_cores": os.cpu_count(), "reference": ref, "port": port, "passes": passes,
           "port_over_reference": {"slice": med("slice"), "c1": med("c1"), "statistic": f"median of {len(passes)} passes"}}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["port_over_reference"]), this)


if __name__ == "__main__":
    main()
