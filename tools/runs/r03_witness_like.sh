#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for h in 0 1; do echo "## option hot = $h"; HOT=$h timeout -k 10 500 python3 tools/runs/r03_witness_like.py 2>&1 | grep -v "amdgpu.ids\|stages of one"; done | tee gpurun_out/r03_witness_like_128.txt
