#!/bin/bash
# round 4: per-stream timeline of the busiest rank's share of n = 2^20 at 8 ranks, run alone on one GPU:
#   2x4 (2^19 points, 4 of 16 windows) and 1x8 (2^20 points, 2 of 16 windows), pipelined
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "19 4 3 a" "20 8 4 b"; do
  set -- $cfg
  d=gpurun_out/prof_share_$4
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python3 tools/bench_shard.py --logn $1 --world $2 --depth $3 --steps 300 > gpurun_out/r04_share_timeline_$4.txt 2> $d.err
  f=$(ls $d/*/*kernel_trace.csv 2>/dev/null || ls $d/*kernel_trace.csv)
  python3 tools/trace_timeline.py $f 60 >> gpurun_out/r04_share_timeline_$4.txt 2>&1
  echo "done $cfg"
done
