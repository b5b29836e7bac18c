#!/bin/bash
# bench.py's N > 1 path rehearsed on the one-GPU box: 2 and 4 ranks share GPU 0, partials exchanged over gloo (MSM_BENCH_BACKEND=gloo);
# the combined point must be the one-GPU bench's (result_x_prefix).  Timings mean nothing here (the ranks share one chip).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_gloo_rehearsal.txt; : > $O
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=1', d['result_x_prefix'], d['config'].get('split'))" >> $O || exit 1
for N in 2 4; do
  MSM_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $N --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/r03_gloo_rehearsal_$N.err | tee gpurun_out/r03_gloo_rehearsal_$N.out | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('N=$N', d['result_x_prefix'], d['config'].get('split'), d['config'].get('exchange'), 'ms/step %.2f' % d['ms_per_step'])" >> $O || { tail -5 gpurun_out/r03_gloo_rehearsal_$N.err; head -c 600 gpurun_out/r03_gloo_rehearsal_$N.out; exit 1; }
done
MSM_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --logn 22 2>/dev/null | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('N=2 logn=22', d['result_x_prefix'], d['config'].get('split'))" >> $O || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --logn 22 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=1 logn=22', d['result_x_prefix'])" >> $O || exit 1
cat $O
