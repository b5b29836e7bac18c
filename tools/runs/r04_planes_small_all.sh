#!/bin/bash
# round 4: the plane kernels at the small end, all four curve / group pairs (reduce_planes 0 | 1): no size may lose
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_planes_small_all.jsonl; : > $out
for cv in bls12381 bn128; do for g in 1 2; do for L in 10 12 14 16; do for rp in 0 1; do
  python3 bench.py --no-cpu-baseline --curve $cv --group $g --logn $L --steps 150 --warmup 15 --opt reduce_planes=$rp >> $out 2>/dev/null
done; done; done; echo "done $cv"; done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], d["config"].get("engine_options"), "c=%d" % d["config"]["window_bits"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", round((d.get("latency") or {}).get("median_ms") or 0, 3))
PY
