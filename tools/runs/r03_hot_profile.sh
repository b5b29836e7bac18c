#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hot -o p -- python3 tools/runs/r03_hot_profile.py > gpurun_out/r03_hot_profile.log 2>&1; echo rc=$?
tail -2 gpurun_out/r03_hot_profile.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_hot/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"].split("(")[0][-70:].ljust(72), r["Calls"].rjust(5), "%9.1f us avg" % (float(r["AverageNs"]) / 1e3), "%6.2f %%" % float(r["Percentage"]))
PY
