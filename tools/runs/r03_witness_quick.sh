#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
HOT=1 timeout -k 10 500 python3 tools/runs/r03_witness_like.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_witness_quick.txt
