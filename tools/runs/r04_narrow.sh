#!/bin/bash
# round 4: scalars narrower than their 32 bytes (uniform below 2^bits), adaptation on (default) and off (--opt narrow=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 16 18 20 22; do for b in 32 64 128 200 253; do for nar in 0 1; do
  python3 bench.py --logn $L --steps 60 --no-cpu-baseline --scalar-bits $b --opt narrow=$nar 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=2^$L bits=$b narrow=$nar', 'c=%d W=%d' % (d['config']['window_bits'], d['config']['windows']), 'ms/MSM %.3f' % d['ms_per_step'], 'one call %.3f' % d['latency']['median_ms'], d['config']['hints_after_timed_region'])"
done; done; done
