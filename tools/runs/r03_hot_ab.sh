#!/bin/bash
# uniform scalars, option hot = 0 against 1 on one box: the hot-bucket path must not cost ordinary calls anything
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for h in 0 1; do for a in "--logn 20" "--logn 23 --steps 20" "--logn 16 --steps 300"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a --opt hot=$h 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a hot=$h'.ljust(40), 'ms/step %.3f  lat %.3f' % (d['ms_per_step'], d['latency']['median_ms']))"; done; done; done | tee gpurun_out/r03_hot_ab.txt
