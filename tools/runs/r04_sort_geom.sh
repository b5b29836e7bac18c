#!/bin/bash
# round 4: the bucket sort's new geometry (sort_geom = 0) against round 3's (= 1) at the sizes it changes: stage times one call at a
# time (stages_ms_one_call_at_a_time of bench.py), ms per MSM pipelined, and the prepared tables whose merged windows it serves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_sort_geom.jsonl; : > $out
for L in 22 23 24 25 26; do
  steps=20; [ $L -ge 25 ] && steps=8
  for sg in 1 0; do
    python3 bench.py --logn $L --steps $steps --warmup 3 --no-cpu-baseline --opt sort_geom=$sg >> $out 2>> gpurun_out/r04_sort_geom.err
    echo "done logn=$L sort_geom=$sg"
  done
done
for L in 20 22; do
  for sg in 1 0; do
    python3 bench.py --logn $L --steps 30 --warmup 3 --no-cpu-baseline --prepared --opt sort_geom=$sg >> $out 2>> gpurun_out/r04_sort_geom.err
    echo "done prepared logn=$L sort_geom=$sg"
  done
done
python3 - <<PY
import json
for ln in open("$out"):
    d = json.loads(ln)
    iso = d.get("stages_ms_one_call_at_a_time") or {}
    sc = sum(iso.get(k, 0) for k in ("histogram", "scan_order", "scatter"))
    print(d["config"]["workload"], "prepared" if "PREPARED" in d["metric"] else "", "c=%d" % d["config"]["window_bits"], "ms/MSM %.3f" % d["ms_per_step"],
          "scatter stage alone %.3f ms" % sc, {k: round(v, 3) for k, v in iso.items() if k in ("histogram", "scan_order", "scatter", "accumulate", "reduce")},
          "frac_scatter_alone", d["roofline_scatter"].get("frac_one_call_at_a_time"))
PY
