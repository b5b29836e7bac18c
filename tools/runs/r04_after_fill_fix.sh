#!/bin/bash
# round 4: after sizing a slot's reduction buffers for both forms -- the driver's form traced again, the GPU suite, the bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
bash tools/runs/r04_fill_drain.sh > /dev/null || exit 1
head -12 $O/r04_fill_drain_a.txt | cut -c1-120; tail -3 $O/r04_fill_drain_a.txt | cut -c1-200
for i in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form ms/step', d['ms_per_step'])"; done
timeout -k 10 300 python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default ms/step', d['ms_per_step'], 'latency', d['latency']['median_ms'])"
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r04_gpu_tests_fill_fix.log 2>&1; echo "tests rc=$?"; tail -2 $O/r04_gpu_tests_fill_fix.log
