#!/bin/bash
# zero_copy = 1 (default) against 0 with calls pipelined, same box, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for z in 0 1; do for a in "--logn 20" "--logn 16 --steps 300"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a --opt zero_copy=$z 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a zero_copy=$z'.ljust(40), 'ms/step %.3f' % d['ms_per_step'])"; done; done; done | tee gpurun_out/r03_zero_copy_pipelined.txt
