#!/bin/bash
# buckets per lane of the reduction (reduce_l) for the shares of a sharded n = 2^20 / 2^22 call: automatic against 16 and 32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_reduce_l_shares.txt; : > $O
for L in 0 16 32; do
  echo "## reduce_l=$L" >> $O
  timeout -k 10 500 python tools/shard_table.py --logn 20 22 --worlds 4 8 --opt reduce_l=$L --budget 0.3 2>&1 | grep -v "^# per-share\|^share" >> $O || exit 1
done
cat $O
