#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q > gpurun_out/r03_gpu_tests_round3_file.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03_gpu_tests_round3_file.log
