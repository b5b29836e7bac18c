#!/bin/bash
# round 4 soak after the plane kernels: the randomised GPU tests under extra seeds (one pytest process)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MSM_TEST_EXTRA_SEEDS=${1:-201,202,203,204,205,206,207,208,209,210}
timeout -k 10 1100 python -u -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -k randomised > gpurun_out/r04_soak_planes.log 2>&1; echo "soak rc=$?"; tail -6 gpurun_out/r04_soak_planes.log
