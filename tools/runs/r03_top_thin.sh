#!/bin/bash
# BN254 with 16-bit windows: the old layout (top window half used) against the even layout over 254 bits (experiment: even_windows = 2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--curve bn128 --logn 19" "--curve bn128 --logn 20" "--curve bn128 --logn 21 --steps 60" "--curve bn128 --logn 22 --steps 30" "--curve bn128 --group 2 --logn 20 --steps 30"; do for ev in 1 2; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a --opt even_windows=$ev 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a even_windows=$ev'.ljust(60), 'c=%d W=%d ms/step %.3f lat %.3f' % (d['config']['window_bits'], d['config']['windows'], d['ms_per_step'], d['latency']['median_ms']), {k: round(v,3) for k,v in d['stages_ms_one_call_at_a_time'].items() if k in ('accumulate','reduce','scatter')}, d['result_x_prefix'])"; done; done | tee gpurun_out/r03_top_thin.txt
