b() { python bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.1f Mpairs/s' % (d['ms_per_step'], d['value']/1e6), d['config'].get('windows'), d.get('stages_ms_one_call_at_a_time') or d.get('stages_ms'))"; }
for r in 0 1; do python tools/bench_shard.py --logn 20 --world 8 --rank $r --depth 4 --steps 40 | cut -c1-70; done
python tools/bench_shard.py --logn 20 --world 4 --depth 2 --steps 40 | cut -c1-70
echo "n=2^20: $(b --logn 20)"
echo "n=2^20 lone: $(b --logn 20 --no-pipeline)"
echo "n=2^16: $(b --logn 16)"
