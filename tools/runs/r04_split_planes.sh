cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 18 20 22; do for sp in 0 1; do
python3 bench.py --no-cpu-baseline --logn $L --steps 30 --warmup 5 --no-pipeline --opt split=$sp 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['config']['engine_options'], 'ms/MSM %.3f' % d['ms_per_step'], 'latency %.3f' % d['latency']['median_ms'])"
done; done
python3 -m pytest tests/test_gpu_round4.py -m gpu -q -k "plain_sums or density" 2>&1 | tail -2
