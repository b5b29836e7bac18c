#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tools/runs/r03_max_call.py > gpurun_out/r03_max_call.txt 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r03_max_call.txt | tail
