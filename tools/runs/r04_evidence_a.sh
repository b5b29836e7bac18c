#!/bin/bash
# round 4 evidence, part A (one gpurun call, ~12 min): GPU tests, smoke, Node golden, bench lines (default, driver form, prepared), rocprofv3
# kernel stats of the same commands, the other configs and the n sweep
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r04_gpu_tests_$tag.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_gpu_tests_$tag.log
timeout -k 10 300 python3 __graft_entry__.py smoke > $O/r04_smoke_$tag.log 2>&1; echo "smoke rc=$?"; tail -1 $O/r04_smoke_$tag.log | cut -c1-200
(cd zprize-wasm-msm_amd/node && timeout -k 10 300 node test_golden.js > ../../$O/r04_node_golden_$tag.log 2>&1; echo "node rc=$?"; tail -2 ../../$O/r04_node_golden_$tag.log)
timeout -k 10 400 python3 bench.py > $O/r04_bench_$tag.json 2> $O/r04_bench_$tag.err; echo "bench rc=$?"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_${tag}_driver_form.json 2>/dev/null
timeout -k 10 400 python3 bench.py --prepared --no-cpu-baseline > $O/r04_bench_${tag}_prepared.json 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_pipelined -o p -- python3 bench.py --no-cpu-baseline > $O/r04_bench_under_rocprof_${tag}_pipelined.json 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_one -o p -- python3 bench.py --no-cpu-baseline --no-pipeline --steps 30 > $O/r04_bench_under_rocprof_${tag}_one_call.json 2>/dev/null
cp $O/prof_${tag}_pipelined/p_kernel_stats.csv $O/r04_bench_kernel_stats_${tag}_pipelined.csv
cp $O/prof_${tag}_one/p_kernel_stats.csv $O/r04_bench_kernel_stats_${tag}_one_call_at_a_time.csv
: > $O/r04_bench_other_configs_$tag.jsonl
run() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" >> $O/r04_bench_other_configs_$tag.jsonl 2>/dev/null || echo "{\"failed\": \"$*\"}" >> $O/r04_bench_other_configs_$tag.jsonl; }
run --logn 16 --steps 200; run --logn 17 --steps 200; run --logn 18 --steps 100; run --logn 20 --steps 100; run --logn 22 --steps 30; run --logn 23 --steps 20
run --logn 24 --steps 10 --warmup 2; run --logn 25 --steps 6 --warmup 2; run --logn 26 --steps 4 --warmup 1
run --group 2 --logn 20 --steps 30; run --curve bn128 --logn 24 --steps 10 --warmup 2; run --curve bn128 --group 2 --logn 20 --steps 30
run --curve bn128 --logn 16 --steps 200; run --curve bn128 --logn 18 --steps 100; run --curve bn128 --logn 20 --steps 100; run --curve bn128 --logn 22 --steps 30
run --prepared --logn 16 --steps 200; run --prepared --logn 18 --steps 100; run --prepared --logn 20 --steps 100; run --prepared --logn 22 --steps 30
run --prepared --group 2 --logn 18 --steps 50; run --prepared --curve bn128 --logn 20 --steps 100
run --host-buffers --steps 20; run --host-buffers --resident-bases --steps 20
MSM_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 50 > $O/r04_bench_forced_dist_cabi_world1.json 2>/dev/null; echo "forced dist rc=$?"
MSM_BENCH_BACKEND=gloo MSM_AMD_RCCL_LIB=$GRAFT_REPO_ROOT/tests/host/libfake_rccl.so timeout -k 10 300 python3 bench.py --gpus 2 --no-cpu-baseline --steps 50 > $O/r04_bench_two_ranks_one_gpu_cabi.json 2>/dev/null; echo "two ranks on one GPU rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/r04_bench_$tag.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","mode")}); print(d["latency"]); print({k:d["roofline"][k] for k in ("bound","achieved","frac","traffic")}); print({k:d["roofline_scatter"][k] for k in ("avg_ms_one_call_at_a_time","frac_one_call_at_a_time","traffic","traffic_frac_of_peak","ceiling")}); print(d["stages_ms_one_call_at_a_time"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["matches_gpu_result"]); print(d["config"]["hints_after_timed_region"])
for f in ("r04_bench_${tag}_driver_form.json","r04_bench_${tag}_prepared.json","r04_bench_forced_dist_cabi_world1.json","r04_bench_two_ranks_one_gpu_cabi.json"):
    try:
        x=json.loads(open("$O/"+f).read().strip().splitlines()[-1]); print(f, "ms/step=%.3f"%x["ms_per_step"], x["config"].get("split"), x["config"].get("exchange"), x["config"].get("rccl_ranks"))
    except Exception as e: print(f, "FAILED", e)
for ln in open("$O/r04_bench_other_configs_$tag.jsonl"):
    x=json.loads(ln)
    if "failed" in x: print(x); continue
    iso = x.get("stages_ms_one_call_at_a_time") or {}
    print(x["config"]["workload"], "prepared" if isinstance(x["config"]["bases"], dict) else "", "c=",x["config"]["window_bits"], "d=",x["config"]["calls_in_flight"], "ms/step=%.3f"%x["ms_per_step"], "%.3e pairs/s"%x["value"], "lat=%s"%(("%.3f"%x["latency"]["median_ms"]) if x.get("latency") else "-"), "frac=%.3f"%x["roofline"]["frac"], "scatter_alone=%.3f ms frac %.3f"%(sum(iso.get(k,0) for k in ("histogram","scan_order","scatter")), x["roofline_scatter"].get("frac_one_call_at_a_time") or 0), x["data"][-34:])
PY
