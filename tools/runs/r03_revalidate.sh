#!/bin/bash
# after the refactor (engine.hip split into stages and translation units): the whole GPU suite, smoke, node, two bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r03_gpu_tests_final.log 2>&1; echo "tests rc=$?"; tail -3 $O/r03_gpu_tests_final.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke_final.log 2>&1; echo "smoke rc=$?"
timeout -k 10 200 node zprize-wasm-msm_amd/node/test_golden.js > $O/r03_node_golden_final.log 2>&1; echo "node rc=$?"; tail -1 $O/r03_node_golden_final.log
for a in "" "--prepared" "--logn 16 --steps 200" "--logn 23 --steps 20" "--group 2 --logn 18 --steps 50"; do timeout -k 10 300 python3 bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a'.ljust(30), 'c=%d W=%d d=%d ms/step %.3f lat %.3f %s' % (d['config']['window_bits'], d['config']['windows'], d['config']['calls_in_flight'], d['ms_per_step'], d['latency']['median_ms'], d['result_x_prefix']))"; done
