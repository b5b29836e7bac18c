#!/bin/bash
# round 4: witness-like scalar mixes at n = 2^20 (tools/runs/r04_witness_profile.py) with the reduction chosen by the density the last
# call reported (default) against k_bucket_reduce always (reduce_planes=0), two and three calls in flight
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_witness_like_sparse.txt; : > $out
for c in 0 1 2 3 4; do
  for opt in reduce_planes=0 reduce_planes=1; do
    echo "-- $opt" >> $out
    python3 tools/runs/r04_witness_profile.py $c 80 $opt 2>/dev/null | cut -c1-400 >> $out
  done
  echo "done $c"
done
