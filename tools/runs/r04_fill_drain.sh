#!/bin/bash
# round 4: kernel timeline of the driver's form (20 timed calls after 5 warmup calls), where the 20-step figure loses to the 100-step one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
for v in a b; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_fill_$v -o p -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_fill_drain_bench_$v.json 2>/dev/null || exit 1
f=$(ls $O/prof_fill_$v/*/*kernel_trace.csv 2>/dev/null || ls $O/prof_fill_$v/*kernel_trace.csv)
python3 tools/fill_drain_trace.py $f > $O/r04_fill_drain_$v.txt 2>&1
python3 -c "
import json; d=json.loads(open('$O/r04_fill_drain_bench_$v.json').read().strip().splitlines()[-1]); print('bench under rocprof: ms/step', d['ms_per_step'])" >> $O/r04_fill_drain_$v.txt
rm -rf $O/prof_fill_$v
done
cat $O/r04_fill_drain_a.txt
