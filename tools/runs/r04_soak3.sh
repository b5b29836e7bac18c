#!/bin/bash
# round 4 soak at the final tree (deferred reduction in): the randomised GPU tests and the random-order test under 40 more seeds (one pytest process)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MSM_TEST_EXTRA_SEEDS=$(seq -s, 701 740)
timeout -k 10 1000 python -u -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -k "randomised or random_orders" > gpurun_out/r04_soak_final_seeds_701_740.log 2>&1; echo "soak rc=$?"; tail -4 gpurun_out/r04_soak_final_seeds_701_740.log
