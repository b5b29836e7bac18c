#!/bin/bash
# round 4, the head of the round on a fresh box -- GPU suite, smoke, the default bench line and the driver's form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_head.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/r04_gpu_tests_head.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 __graft_entry__.py smoke > $O/r04_smoke_head.log 2>&1 && echo "smoke ok" && \
timeout -k 10 400 python3 bench.py > $O/r04_bench_head.json 2> $O/r04_bench_head.err && echo "bench ok" && \
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_head_driver_form.json 2>/dev/null && \
python3 - <<PY
import json
for f in ("r04_bench_head.json","r04_bench_head_driver_form.json"):
    d=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["latency"]["median_ms"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("matches_gpu_result"))
PY
