#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in 1; do
  d=gpurun_out/prof_wit$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 tools/runs/r04_witness_profile.py $c 60 > gpurun_out/r04_witness_profile_$c.txt 2> $d.err
  python3 - <<PY >> gpurun_out/r04_witness_profile_$c.txt
import csv,glob
f=glob.glob("$d/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.1f us" % (float(r["AverageNs"])/1e3)).rjust(12), r["Percentage"])
PY
  echo "done $c"
done
for c in 1 2 3; do python3 tools/runs/r04_witness_profile.py $c 80 depth=3 2>/dev/null | cut -c1-60; done
