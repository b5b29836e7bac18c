#!/bin/bash
# the host combine with the ADX product (field_host_adx.hpp): stage time and the latency of one call, a few workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "--logn 20" "--logn 16" "--group 2 --logn 20 --steps 30" "--curve bn128 --logn 20" "--prepared --logn 20"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pipeline --steps 60 --warmup 10 $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a'.ljust(36), 'latency median %.3f ms   host_combine %.3f ms' % (d['latency']['median_ms'], d['stages_ms']['host_combine']))"; done | tee gpurun_out/r03_host_adx.txt
