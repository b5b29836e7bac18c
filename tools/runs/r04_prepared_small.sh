#!/bin/bash
# round 4: prepared tables at small n -- plane kernels on the merged windows (reduce_planes 0 | 1), three against four calls in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_prepared_small.jsonl; : > $out
for L in 16 18 20; do
  for rp in 0 1; do
    for d in 2 3 4; do
      python3 bench.py --logn $L --steps 150 --warmup 10 --no-cpu-baseline --prepared --depth $d --opt reduce_planes=$rp >> $out 2>/dev/null
    done
  done
  echo "done $L"
done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "PREPARED", d["config"].get("engine_options"), "c=%d" % d["config"]["window_bits"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", round((d.get("latency") or {}).get("median_ms") or 0, 3))
PY
