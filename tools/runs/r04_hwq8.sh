#!/bin/bash
# round 4: eight hardware queues (GPU_MAX_HW_QUEUES=8) with three to six small calls in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_hwq8.txt; : > $out
for q in 4 8; do
  for L in 16 18; do
    for d in 3 4 6; do
      GPU_MAX_HW_QUEUES=$q python3 bench.py --logn $L --steps 200 --warmup 20 --no-cpu-baseline --depth $d 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q', d['config']['workload'], 'in flight', d['config']['calls_in_flight'], 'ms/MSM %.3f' % d['ms_per_step'])" >> $out
    done
  done
  for d in 3 4 6; do GPU_MAX_HW_QUEUES=$q python3 tools/bench_shard.py --logn 19 --world 4 --depth $d --steps 300 2>/dev/null | cut -c1-50 | sed "s/^/queues $q /" >> $out; done
done
cat $out
