#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_lifecycle.py -m gpu -x -q > gpurun_out/r03_lifecycle.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r03_lifecycle.log
