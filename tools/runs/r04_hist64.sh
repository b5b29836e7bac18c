#!/bin/bash
# round 4: one-wave workgroups for the histogram of a pipelined call (MSM_AMD_HIST_BLOCK=64), and the hidden reduction's L, at n = 2^20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
one() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hist=%s' % os.environ.get('MSM_AMD_HIST_BLOCK'), '$*', 'ms/step %.4f' % d['ms_per_step'], 'latency %.3f' % d['latency']['median_ms'], d.get('cpu_baseline'))"; }
timeout -k 10 200 env MSM_AMD_HIST_BLOCK=64 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py -m gpu -x -q -k "golden or deferred or randomised" > $O/r04_hist64_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r04_hist64_tests.log
for i in 1 2; do
one; MSM_AMD_HIST_BLOCK=64 one
one --steps 20 --warmup 5; MSM_AMD_HIST_BLOCK=64 one --steps 20 --warmup 5
done
one --opt reduce_l=16; one --opt reduce_l=64
MSM_AMD_HIST_BLOCK=64 one --curve bn128; one --curve bn128
MSM_AMD_HIST_BLOCK=64 one --logn 22 --steps 30; one --logn 22 --steps 30
export MSM_AMD_HIST_BLOCK=64
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_h64 -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
f=$(ls $O/prof_h64/*/*kernel_trace.csv 2>/dev/null || ls $O/prof_h64/*kernel_trace.csv)
python3 tools/fill_drain_trace.py $f > $O/r04_hist64_trace.txt 2>&1; rm -rf $O/prof_h64
grep -n "call 19" -A 12 $O/r04_hist64_trace.txt | cut -c1-110; tail -3 $O/r04_hist64_trace.txt | cut -c1-200
