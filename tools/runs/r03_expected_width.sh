#!/bin/bash
# windows tile the expected scalar width (the group order's), an overflow window above; the declared width only after wide scalars were seen
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
{
echo "## state machine (tools/runs/r03_skew_state.py)"
timeout -k 10 300 python3 tools/runs/r03_skew_state.py 2>&1 | grep -v amdgpu.ids
echo "## bn128 and bls12381, sizes whose window size does not divide 256"
for a in "--curve bn128 --logn 16 --steps 200" "--curve bn128 --logn 18 --steps 200" "--curve bn128 --logn 20" "--curve bn128 --logn 22 --steps 30" "--curve bn128 --logn 24 --steps 10" "--curve bn128 --group 2 --logn 18 --steps 60" "--logn 16 --steps 200" "--logn 18 --steps 200" "--logn 20" "--logn 23 --steps 20" "--logn 24 --steps 10" "--group 2 --logn 18 --steps 60"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a'.ljust(44), 'c=%d W=%d d=%d ms/step %.3f lat %.3f' % (d['config']['window_bits'], d['config']['windows'], d['config']['calls_in_flight'], d['ms_per_step'], d['latency']['median_ms']), {k: round(v,3) for k,v in d['stages_ms_one_call_at_a_time'].items() if k in ('accumulate','reduce','scatter')}, d['result_x_prefix'])"; done
} | tee gpurun_out/r03_expected_width.txt
