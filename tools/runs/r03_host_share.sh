#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 0x7F 0x3F; do echo "## top byte mask $m"; TOP_MASK=$m timeout -k 10 300 python3 tools/runs/r03_host_share.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03_host_share.txt
