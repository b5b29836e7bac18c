#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/runs/r03_s2_state.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_s2_state.txt
