#!/usr/bin/env python3
"""Per-kernel HBM traffic of one bench.py run from the hardware counters (run on the GPU box).

    python3 tools/pmc_per_kernel.py OUT.json [bench.py arguments ...]

Runs `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` as two separate passes (the TCC block cannot hold
both: MI355X_MICROARCH.md, rocprofv3 PMC slots), each around `python3 bench.py --no-cpu-baseline --no-pipeline <args>` (the program
itself after `--`, no shell hop), and writes the mean per launch of every kernel:
    {kernel: {"launches": k, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "hbm_MB_fetch_x2": 2f + w, "hbm_MB_fetch_x1": f + w}}
FETCH_SIZE on gfx950 reports half of the bytes of a wide coalesced streaming read (the guide's correction: double it); on narrower
or gathered accesses the factor is uncalibrated and lies between 1 and 2, so both sums are given.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, bench_args, tag):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{counter}")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--",
           "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-pipeline"] + bench_args
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {out}")
    acc = {}
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        s = acc.setdefault(name, [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    # launch durations of the same pass (kernel trace)
    dur = {}
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            try:
                d = dur.setdefault(r["Kernel_Name"], [0, 0.0])
                d[0] += 1
                d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
    DUR.update({k: v[1] / v[0] for k, v in dur.items()})
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


DUR = {}   # kernel -> mean launch duration in ns (of the last pass that saw it)


def short(name):
    name = name.replace("void msm::", "").replace("msm::", "")
    return name.split("(")[0]


def main():
    out_json = sys.argv[1]
    bench_args = sys.argv[2:] or ["--steps", "5", "--warmup", "2"]
    tag = os.path.splitext(os.path.basename(out_json))[0]
    fetch = one_pass("FETCH_SIZE", bench_args, tag)
    write = one_pass("WRITE_SIZE", bench_args, tag)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if "k_spin" in k or "k_gen_points" in k:
            continue
        n, f = fetch.get(k, (0, 0.0))
        _, w = write.get(k, (0, 0.0))
        res[short(k)] = {"launches": n, "FETCH_SIZE_KB_raw": round(f, 2), "WRITE_SIZE_KB": round(w, 2),
                         "hbm_MB_fetch_x2": round((2 * f + w) * 1024 / 1e6, 2), "hbm_MB_fetch_x1": round((f + w) * 1024 / 1e6, 2)}
        if DUR.get(k):
            us = DUR[k] / 1e3
            res[short(k)]["avg_us"] = round(us, 2)
            res[short(k)]["GBps_on_moved_bytes_fetch_x2"] = round((2 * f + w) * 1024 / (us * 1e-6) / 1e9, 1)
    doc = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --no-cpu-baseline --no-pipeline " + " ".join(bench_args),
           "units": "mean per launch; KB as rocprofv3 reports (x1024 bytes); MB = 1e6 bytes",
           "kernels": res}
    json.dump(doc, open(out_json, "w"), indent=1)
    for k, v in res.items():
        print(f"{k[:70]:70s} {v['launches']:4d}  fetch_raw {v['FETCH_SIZE_KB_raw'] / 1024:9.2f} MiB  write {v['WRITE_SIZE_KB'] / 1024:9.2f} MiB")


if __name__ == "__main__":
    main()
