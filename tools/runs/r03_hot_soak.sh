#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python3 tools/runs/r03_hot_soak.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_hot_soak.txt; exit ${PIPESTATUS[0]}
