#!/bin/bash
# round 4: k_bucket_planes at window sizes above 16 bits and on prepared tables (reduce_planes = 0 | 1 | 2), after the rule was narrowed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_planes3.jsonl; : > $out
run() { python3 bench.py --no-cpu-baseline "$@" >> $out 2>> gpurun_out/r04_planes3.err; echo "done $*"; }
for rp in 0 1 2; do
  run --logn 23 --steps 12 --warmup 3 --opt reduce_planes=$rp
  run --logn 20 --steps 60 --warmup 5 --prepared --opt reduce_planes=$rp
  run --logn 22 --steps 20 --warmup 3 --prepared --opt reduce_planes=$rp
  run --logn 20 --steps 60 --warmup 5 --prepared --curve bn128 --opt reduce_planes=$rp
  run --logn 24 --steps 8 --warmup 2 --opt reduce_planes=$rp
done
run --logn 20 --steps 30 --warmup 5 --no-pipeline
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "PREP" if "PREPARED" in d["metric"] else "", d["config"].get("engine_options"), "c=%d" % d["config"]["window_bits"], "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", round((d.get("latency") or {}).get("median_ms") or 0, 3), {k: round(v, 3) for k, v in (d.get("stages_ms_one_call_at_a_time") or {}).items() if k in ("histogram", "reduce")})
PY
