#!/bin/bash
# round 4: the bucket reduction as plain sums (option reduce_planes = 1, k_bucket_planes) against k_bucket_reduce (= 0):
# ms per MSM pipelined at n = 2^16 ... 2^20, one call at a time at 2^20, and the shares of n = 2^20 at 8 ranks
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_planes.jsonl; : > $out
for L in 16 17 18 20; do
  for rp in 0 1; do
    python3 bench.py --logn $L --steps 200 --warmup 20 --no-cpu-baseline --opt reduce_planes=$rp >> $out 2>> gpurun_out/r04_planes.err
    echo "done logn=$L planes=$rp"
  done
done
for rp in 0 1; do
  python3 bench.py --logn 20 --steps 60 --warmup 5 --no-cpu-baseline --no-pipeline --opt reduce_planes=$rp >> $out 2>> gpurun_out/r04_planes.err
done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    print(d["config"]["workload"], "c=%d" % d["config"]["window_bits"], d["config"].get("engine_options"), "ms/MSM %.3f" % d["ms_per_step"], "latency", (d.get("latency") or {}).get("median_ms"), {k: round(v, 3) for k, v in (d.get("stages_ms_one_call_at_a_time") or {}).items() if k in ("accumulate", "reduce")})
PY
for cfg in "19 4 3" "20 8 4" "17 1 3"; do
  set -- $cfg
  for rp in 0 1; do
    python3 tools/bench_shard.py --logn $1 --world $2 --depth $3 --steps 300 --opt reduce_planes=$rp 2>/dev/null | cut -c1-120
  done
done
echo "# buckets per lane (reduce_l) with the plane kernels: n = 2^16 and 2^20, pipelined | one call at a time"
out2=gpurun_out/r04_planes_l.jsonl; : > $out2
for L in 16 20; do
  for rl in 2 4 8 16 32; do
    python3 bench.py --logn $L --steps 150 --warmup 20 --no-cpu-baseline --opt reduce_l=$rl >> $out2 2>> gpurun_out/r04_planes.err
    python3 bench.py --logn $L --steps 40 --warmup 5 --no-cpu-baseline --no-pipeline --opt reduce_l=$rl >> $out2 2>> gpurun_out/r04_planes.err
  done
done
python3 - <<PY
import json
for ln in open("$out2"):
    d = json.loads(ln)
    print(d["config"]["workload"], d["config"].get("engine_options"), "in flight", d["config"].get("calls_in_flight"), "ms/MSM %.3f" % d["ms_per_step"], "latency", (d.get("latency") or {}).get("median_ms"))
PY
