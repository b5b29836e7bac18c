#!/bin/bash
# round 4 evidence, part B: hardware counters per kernel on the workloads bench.py quotes them for (separate --pmc passes, program after --),
# per-workload summaries (tools/make_pmc_summaries.py), and the per-rank shard table of BLS12-381 G1 with this round's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
one() {  # key, bench args...
  key=$1; shift
  timeout -k 10 600 python3 tools/pmc_per_kernel.py $O/r04_pmc_fetch_write_$key.json "$@" > $O/r04_pmc_$key.txt 2>&1; echo "pmc $key rc=$?"
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-pipeline "$@" > $O/r04_pmc_geom_$key.json 2>/dev/null
  python3 - <<PY
import json, subprocess, sys
d = json.loads(open("$O/r04_pmc_geom_$key.json").read().strip().splitlines()[-1])
n = 1 << int("$key".split("2p")[1])
ent = int(round(d["roofline"]["hbm"]["algorithmic_bytes_per_launch"] / ({"bls12381_g1": 100, "bn128_g1": 68, "bls12381_g2": 196, "bn128_g2": 132}["_".join("$key".split("_")[:2])])))
subprocess.check_call([sys.executable, "tools/make_pmc_summaries.py", "$key", str(n), str(ent), str(d["config"]["window_bits"]), str(d["config"]["windows"]), "$O/r04_pmc_fetch_write_$key.json"])
PY
}
one bls12381_g1_2p20 --steps 5 --warmup 2
one bls12381_g1_2p24 --logn 24 --steps 3 --warmup 1
one bls12381_g1_2p26 --logn 26 --steps 2 --warmup 1
one bls12381_g2_2p20 --group 2 --steps 4 --warmup 1
one bn128_g1_2p24 --curve bn128 --logn 24 --steps 3 --warmup 1
timeout -k 10 900 python3 tools/shard_table.py --logn 16 18 20 22 24 26 --out $O/r04_shard_per_rank_table.jsonl > $O/r04_shard_per_rank_table.txt 2>&1; echo "shard table rc=$?"; grep "^n=2" $O/r04_shard_per_rank_table.txt | cut -c1-300
