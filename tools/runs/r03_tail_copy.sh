#!/bin/bash
# what lies between the end of the reduction and the end of the call on the GPU side (the copies back to the host): gpu_total minus the stages;
# zero_copy = 1: the reduction writes the partials straight into the pinned host buffer
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--logn 20" "--logn 16" "--curve bn128 --logn 18" "--group 2 --logn 18"; do for z in 0 1 0 1; do timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pipeline --steps 100 --warmup 10 $a --opt zero_copy=$z 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stages_ms']; rest=s['gpu_total']-sum(s[k] for k in ('histogram','scan_order','scatter','accumulate','reduce','to_affine'))
print('$a zero_copy=$z'.ljust(40), 'gpu_total %.3f  after the reduction %.3f ms  reduce %.3f  latency %.3f  %s' % (s['gpu_total'], rest, s['reduce'], d['latency']['median_ms'], d['result_x_prefix']))"; done; done | tee gpurun_out/r03_tail_copy.txt
