#!/usr/bin/env python3
"""Effective clock of every k_accumulate launch of one bench.py run, in launch order: GRBM_GUI_ACTIVE / 8 XCDs / duration from one
`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` pass (run on the GPU box; counter collection serialises kernels, so this is the one-call-at-a-
time form whatever the arguments say).  usage: clock_per_launch.py [bench.py arguments ...]"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "pmc_clock")
cmd = ["rocprofv3", "--kernel-trace", "--pmc", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", out, "-o", "p", "--",
       "python3", os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:]
r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
if r.returncode != 0:
    sys.exit("rocprofv3 failed: " + r.stderr.decode()[-400:])
dur = {}
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Dispatch_Id"]] = (int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"])
rows = []
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE" and "k_accumulate<" in row["Kernel_Name"] and row["Dispatch_Id"] in dur:
            s, e, _ = dur[row["Dispatch_Id"]]
            rows.append((s, e, float(row["Counter_Value"])))
rows.sort()
t0 = rows[0][0]
prev_end = None
for s, e, v in rows:
    gap = "" if prev_end is None else f"  idle before it (no accumulation) {(s - prev_end) / 1e3:9.1f} us"
    print(f"t = {(s - t0) / 1e6:9.2f} ms  k_accumulate {(e - s) / 1e6:6.3f} ms  clock {v / 8 / (e - s):5.3f} GHz{gap}")
    prev_end = e
