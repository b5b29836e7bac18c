#!/bin/bash
# round 4: the deferred reduction (option defer_reduce) -- its test, the driver's form with it on and off, the trace, the default line, the GPU suite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "deferred or plain_sums or density" > $O/r04_defer_test.log 2>&1; rc=$?; echo "defer test rc=$rc"; tail -3 $O/r04_defer_test.log
[ $rc -eq 0 ] || { tail -40 $O/r04_defer_test.log; exit $rc; }
one() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step %.4f' % d['ms_per_step'], 'latency', d['latency']['median_ms'] if d.get('latency') else None)"; }
for i in 1 2 3; do one --gpus 1 --steps 20 --warmup 5; done
for i in 1 2 3; do one --gpus 1 --steps 20 --warmup 5 --opt defer_reduce=0; done
one; one --opt defer_reduce=0
one --logn 22 --steps 20 --warmup 5; one --logn 22 --steps 20 --warmup 5 --opt defer_reduce=0
one --curve bn128 --steps 20 --warmup 5; one --curve bn128 --steps 20 --warmup 5 --opt defer_reduce=0
one --group 2 --steps 20 --warmup 5; one --group 2 --steps 20 --warmup 5 --opt defer_reduce=0
one --prepared --steps 20 --warmup 5; one --prepared --steps 20 --warmup 5 --opt defer_reduce=0
bash tools/runs/r04_fill_drain.sh > /dev/null && { head -3 $O/r04_fill_drain_a.txt | cut -c1-150; tail -3 $O/r04_fill_drain_a.txt | cut -c1-200; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_defer.log 2>&1; echo "tests rc=$?"; tail -2 $O/r04_gpu_tests_defer.log
