#!/bin/bash
# round 4: does the clock explain the slower accumulations at the head of a timed region?  per-launch effective clock, one call at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/clock_per_launch.py --no-pipeline --steps 20 --warmup 5 > gpurun_out/r04_clock_per_launch.txt 2>&1; echo rc=$?
rm -rf gpurun_out/pmc_clock; cat gpurun_out/r04_clock_per_launch.txt | cut -c1-150
