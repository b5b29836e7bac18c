#!/usr/bin/env python3
"""Hardware counters of the engine's kernels over one bench.py run, one rocprofv3 pass per counter group (run on the GPU box).

    python3 tools/pmc_kernel_counters.py OUT.json "CTR_A,CTR_B;CTR_C" [bench.py arguments ...]

Counter groups are separated by ';' (one pass each: a pass holds what one hardware block can count at once), counters of a group by
','.  Each pass is `rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --no-cpu-baseline --no-pipeline <args>` (program after
`--`, no shell hop, no other trace domain).  Output: mean per launch of every counter for every kernel."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(group, bench_args, tag, idx):
    out = os.path.join(ROOT, "gpurun_out", f"pmcx_{tag}_{idx}")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + group + ["--output-format", "csv", "-d", out, "-o", "p", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pipeline"] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
    if r.returncode != 0:
        print(f"pass {group}: rocprofv3 failed: {r.stderr.decode()[-400:]}", file=sys.stderr)
        return {}
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"], r["Counter_Name"])
            s = acc.setdefault(k, [0, 0.0])
            s[0] += 1
            s[1] += float(r["Counter_Value"])
    # launch durations of the same pass (kernel trace): the effective clock of a kernel is GRBM_GUI_ACTIVE / 8 XCDs / duration
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                k = (r["Kernel_Name"], "duration_ns")
                s = acc.setdefault(k, [0, 0.0])
                s[0] += 1
                s[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


def short(name):
    return name.replace("void msm::", "").replace("msm::", "").split("(")[0]


def main():
    out_json, groups = sys.argv[1], [g.split(",") for g in sys.argv[2].split(";") if g]
    bench_args = sys.argv[3:] or ["--steps", "5", "--warmup", "2"]
    tag = os.path.splitext(os.path.basename(out_json))[0]
    res = {}
    for i, g in enumerate(groups):
        for (kern, ctr), (n, mean) in one_pass(g, bench_args, tag, i).items():
            if "k_spin" in kern or "k_gen_points" in kern:
                continue
            d = res.setdefault(short(kern), {})
            d["launches"] = n
            d[ctr] = mean
    doc = {"command": "rocprofv3 --kernel-trace --pmc <group> (one pass per group) -- python3 bench.py --no-cpu-baseline --no-pipeline " + " ".join(bench_args),
           "groups": groups, "units": "mean per launch, as rocprofv3 reports", "kernels": res}
    json.dump(doc, open(out_json, "w"), indent=1)
    for k, v in res.items():
        if "GRBM_GUI_ACTIVE" in v and v.get("duration_ns"):
            v["effective_clock_GHz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / v["duration_ns"]
    json.dump(doc, open(out_json, "w"), indent=1)
    for k, v in res.items():
        if "accumulate" in k:
            print(k[:60], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
