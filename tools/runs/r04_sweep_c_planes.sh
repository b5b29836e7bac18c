#!/bin/bash
# round 4: window size at small n again, now that the reduction is cheaper (k_bucket_planes): ms per MSM pipelined, latency
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_sweep_c_planes.jsonl; : > $out
for L in 15 16 17 18 19; do
  for c in 0 13 14 15 16; do
    python3 bench.py --logn $L --steps 150 --warmup 20 --no-cpu-baseline --window-bits $c >> $out 2>> gpurun_out/r04_sweep_c_planes.err
  done
  echo "done $L"
done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "c=%d" % d["config"]["window_bits"], "windows", d["config"]["windows"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", round((d.get("latency") or {}).get("median_ms") or 0, 3))
PY
