#!/bin/bash
# round 3, GPU run 11: segment length covers twice the mean uncapped -- BN254 (254-bit order) lines, headline unchanged?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "scalar_widths or all_equal or adversarial or several_lanes or randomised or golden" > gpurun_out/r03_gpu_tests_run11.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_gpu_tests_run11.log
for a in "--curve bn128 --group 2 --logn 20 --steps 30" "--curve bn128 --logn 20 --steps 100" "--curve bn128 --logn 18 --steps 100" "--curve bn128 --group 2 --logn 18 --steps 50" "" "--group 2 --logn 20 --steps 30" "--curve bn128 --logn 24 --steps 10 --warmup 2" "--logn 22 --steps 30"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); st=d['stages_ms_one_call_at_a_time'] or {}
print('$a'.ljust(46), 'c=%d W=%d d=%d ms/step %.3f lat %.3f  alone acc %.3f red %.3f' % (d['config']['window_bits'], d['config']['windows'], d['config']['calls_in_flight'], d['ms_per_step'], d['latency']['median_ms'], st.get('accumulate',0), st.get('reduce',0)))"; done
