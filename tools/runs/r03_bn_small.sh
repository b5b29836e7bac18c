#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--curve bn128 --logn 16 --steps 200" "--curve bn128 --logn 17 --steps 200" "--curve bn128 --logn 18 --steps 200" "--curve bn128 --logn 20" "--curve bn128 --logn 22 --steps 30" "--curve bn128 --logn 24 --steps 10"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a'.ljust(40), 'c=%d W=%d d=%d ms/step %.3f lat %.3f' % (d['config']['window_bits'], d['config']['windows'], d['config']['calls_in_flight'], d['ms_per_step'], d['latency']['median_ms']), {k: round(v,3) for k,v in d['stages_ms_one_call_at_a_time'].items() if k in ('accumulate','reduce','scatter')})"; done | tee gpurun_out/r03_bn_small.txt
