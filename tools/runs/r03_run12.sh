#!/bin/bash
# round 3, GPU run 12: skewed inputs (256-bit scalars: carry window) with the even layout switched on by the skew hint
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_round3.py -m gpu -x -q -k "scalar_widths or all_equal or adversarial or several_lanes or randomised or golden or digits or regrow or histogram" > gpurun_out/r03_gpu_tests_run12.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_gpu_tests_run12.log
for a in "--bits 256" "--bits 255" "--bits 250" "--same" "--bits 256 --opt even_windows=0"; do timeout -k 10 200 python3 tools/bench_shard.py --logn 20 --world 1 --rank 0 --depth 2 --steps 60 $a 2>/dev/null | cut -c1-200; done
for a in "--bits 256" "--bits 255"; do timeout -k 10 200 python3 tools/bench_shard.py --logn 18 --world 1 --rank 0 --depth 3 --steps 100 $a 2>/dev/null | cut -c1-200; done
