#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 tools/runs/r03_prepared_skew.py > gpurun_out/r03_prepared_skew.txt 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r03_prepared_skew.txt | tail -20
