#!/bin/bash
# soak: the two randomised GPU tests under extra seeds (one pytest process), beyond the committed ones
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MSM_TEST_EXTRA_SEEDS=${1:-101,102,103,104,105,106,107,108,109,110,111,112}
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -k randomised > gpurun_out/r03_soak.log 2>&1; echo "soak rc=$?"; tail -15 gpurun_out/r03_soak.log
