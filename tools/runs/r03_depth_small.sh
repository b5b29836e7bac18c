#!/bin/bash
# calls in flight for small calls: 3 (what bench.py picks) against 4, 5, 6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--logn 16 --steps 300" "--logn 17 --steps 300" "--logn 18 --steps 200" "--curve bn128 --logn 16 --steps 300" "--curve bn128 --logn 18 --steps 200"; do for d in 3 4 5 6; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a --depth $d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a depth $d'.ljust(50), 'ms/step %.3f' % d['ms_per_step'])"; done; done | tee gpurun_out/r03_depth_small.txt
