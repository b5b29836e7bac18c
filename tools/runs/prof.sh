#!/bin/bash
# usage: tools/runs/prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ + summary on stdout
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv", recursive=True)
if not f: print("no stats csv"); raise SystemExit(1)
for r in list(csv.DictReader(open(f[0])))[:24]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(6), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(12), r["Percentage"])
PY
