#!/bin/bash
# round 4: three calls in flight at n = 2^20 with 4 / 6 / 8 hardware queues, and the kernel timeline of the default (4) to see why it is slow
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
one() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues=%s' % os.environ.get('GPU_MAX_HW_QUEUES'), '$*', 'ms/step %.4f' % d['ms_per_step'])"; }
for q in 4 6 8; do export GPU_MAX_HW_QUEUES=$q; one --depth 3; one --depth 2; one --depth 3 --steps 20 --warmup 5; done
unset GPU_MAX_HW_QUEUES
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_d3 -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --depth 3 --no-cpu-baseline > /dev/null 2>&1
f=$(ls $O/prof_d3/*/*kernel_trace.csv 2>/dev/null || ls $O/prof_d3/*kernel_trace.csv)
python3 tools/fill_drain_trace.py $f 5 20 > $O/r04_depth3_trace.txt 2>&1; rm -rf $O/prof_d3
sed -n 1,45p $O/r04_depth3_trace.txt | cut -c1-110
