#!/bin/bash
# round 4: calls in flight at small n again, with the plane kernels (bench.py --depth)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_depth_planes.jsonl; : > $out
for L in 15 16 17 18 19; do
  for d in 2 3 4 5; do
    python3 bench.py --logn $L --steps 200 --warmup 20 --no-cpu-baseline --depth $d >> $out 2>> gpurun_out/r04_depth_planes.err
  done
  echo "done $L"
done
python3 bench.py --no-cpu-baseline --steps 100 >> $out 2>/dev/null
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "c=%d" % d["config"]["window_bits"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "hist stage", round((d.get("stages_ms_one_call_at_a_time") or {}).get("histogram", 0), 3))
PY
