#!/bin/bash
# round 4: the sort stream at the greatest HIP priority (its kernels are starved beside an accumulation: k_tile_hist 26 us alone, 620-700 us
# there, and the next accumulation waits for them ~80 us per call) -- and three calls in flight on top of the deferred reduction
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S=%s A=%s' % (os.environ.get('MSM_AMD_S_PRIORITY'), os.environ.get('MSM_AMD_A_PRIORITY')), '$*', 'ms/step %.4f' % d['ms_per_step'], 'latency', d['latency']['median_ms'] if d.get('latency') else None)"; }
for i in 1 2; do
one --steps 20 --warmup 5; one
MSM_AMD_S_PRIORITY=-1 one --steps 20 --warmup 5; MSM_AMD_S_PRIORITY=-1 one
MSM_AMD_S_PRIORITY=1 one --steps 20 --warmup 5
MSM_AMD_A_PRIORITY=1 one --steps 20 --warmup 5; MSM_AMD_A_PRIORITY=1 one
one --depth 3 --steps 20 --warmup 5; one --depth 3
done
MSM_AMD_S_PRIORITY=-1 one --logn 16 --steps 200; one --logn 16 --steps 200
MSM_AMD_S_PRIORITY=-1 one --logn 18 --steps 100; one --logn 18 --steps 100
MSM_AMD_S_PRIORITY=-1 one --logn 24 --steps 10 --warmup 2; one --logn 24 --steps 10 --warmup 2
