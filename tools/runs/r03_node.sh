#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 node zprize-wasm-msm_amd/node/test_golden.js > gpurun_out/r03_node_golden_final.log 2>&1; echo "node rc=$?"; tail -3 gpurun_out/r03_node_golden_final.log
