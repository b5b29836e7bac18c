#!/bin/bash
# round 4: the deferred reduction on G2 calls (defer_reduce = 2) against the default (G1 only), three runs each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step %.4f' % d['ms_per_step'], 'latency', d['latency']['median_ms'] if d.get('latency') else None)"; }
for i in 1 2 3; do
one --group 2 --steps 20 --warmup 5; one --group 2 --steps 20 --warmup 5 --opt defer_reduce=2
one --group 2 --curve bn128 --steps 20 --warmup 5; one --group 2 --curve bn128 --steps 20 --warmup 5 --opt defer_reduce=2
done
one --group 2 --steps 60; one --group 2 --steps 60 --opt defer_reduce=2
