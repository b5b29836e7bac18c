#!/bin/bash
# round 3 evidence, part B: hardware counters per kernel (separate --pmc passes), the per-rank shard table
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python3 tools/pmc_per_kernel.py $O/r03_pmc_fetch_write_per_kernel_$tag.json --steps 5 --warmup 2 > $O/r03_pmc_$tag.txt 2>&1
timeout -k 10 500 python3 tools/pmc_kernel_counters.py $O/r03_pmc_sq_$tag.json "GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY;TCC_HIT_sum,TCC_MISS_sum,TCP_TCC_READ_REQ_sum" --steps 5 --warmup 2 >> $O/r03_pmc_$tag.txt 2>&1
timeout -k 10 500 python3 tools/pmc_per_kernel.py $O/r03_pmc_fetch_write_per_kernel_2p24_$tag.json --logn 24 --steps 3 --warmup 1 >> $O/r03_pmc_$tag.txt 2>&1
cut -c1-260 $O/r03_pmc_$tag.txt
timeout -k 10 560 python3 tools/shard_table.py --out $O/r03_shard_per_rank_table.jsonl > $O/r03_shard_per_rank_table.txt 2>&1; echo "table rc=$?"; head -28 $O/r03_shard_per_rank_table.txt | cut -c1-250
