#!/bin/bash
# round 4: k_bucket_planes under its final rule (reduce_planes = 1) against k_bucket_reduce (= 0) and against planes everywhere (= 2):
# larger n, the other curve and group, shares
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_planes2.jsonl; : > $out
run() { python3 bench.py --no-cpu-baseline "$@" >> $out 2>> gpurun_out/r04_planes2.err; echo "done $*"; }
for rp in 0 1; do
  for L in 14 15 16 18 19; do run --logn $L --steps 150 --warmup 20 --opt reduce_planes=$rp; done
done
for rp in 0 2; do
  run --logn 22 --steps 20 --warmup 3 --opt reduce_planes=$rp
  run --logn 24 --steps 10 --warmup 2 --opt reduce_planes=$rp
  run --logn 22 --steps 10 --warmup 2 --no-pipeline --opt reduce_planes=$rp
  run --logn 24 --steps 6 --warmup 2 --no-pipeline --opt reduce_planes=$rp
  run --logn 20 --steps 30 --warmup 3 --prepared --opt reduce_planes=$rp
  run --logn 18 --steps 60 --warmup 5 --group 2 --opt reduce_planes=$rp
  run --logn 18 --steps 30 --warmup 5 --group 2 --no-pipeline --opt reduce_planes=$rp
  run --logn 20 --steps 60 --warmup 5 --curve bn128 --opt reduce_planes=$rp
  run --logn 20 --steps 30 --warmup 5 --curve bn128 --no-pipeline --opt reduce_planes=$rp
  run --logn 16 --steps 100 --warmup 5 --curve bn128 --opt reduce_planes=$rp
done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "PREP" if "PREPARED" in d["metric"] else "", d["config"].get("engine_options"), "c=%d" % d["config"]["window_bits"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", round((d.get("latency") or {}).get("median_ms") or 0, 3))
PY
