#!/bin/bash
# a lone call reduced in two window groups, high windows first, their Horner part on the host while the GPU finishes the low ones
# (option split = 1) against the default (one group): latency of one call at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_split_early.txt; : > $O
for cfg in "--logn 20" "--logn 22 --steps 30" "--logn 18" "--group 2 --logn 20 --steps 30" "--curve bn128 --logn 20"; do
for sp in 0 1; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pipeline --steps 60 --warmup 10 $cfg --opt split=$sp 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg split=$sp'.ljust(44), 'ms/step %.3f  latency median %.3f min %.3f' % (d['ms_per_step'], d['latency']['median_ms'], d['latency']['min_ms']), {k: round(v, 3) for k, v in d['stages_ms'].items()}, d['result_x_prefix'])" >> $O || exit 1
done; done
cat $O
