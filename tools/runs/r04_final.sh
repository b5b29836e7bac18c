#!/bin/bash
# round 4, final tree: the random-order test of the deferred reduction under 40 extra seeds, then the evidence pass (part A) as v7
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
MSM_TEST_EXTRA_SEEDS=$(seq -s, 601 640) timeout -k 10 500 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "random_orders" > $O/r04_soak_defer_orders_seeds_601_640.log 2>&1; rc=$?; echo "order soak rc=$rc"; tail -2 $O/r04_soak_defer_orders_seeds_601_640.log
[ $rc -eq 0 ] || { tail -40 $O/r04_soak_defer_orders_seeds_601_640.log; exit $rc; }
bash tools/runs/r04_evidence_a.sh v7
