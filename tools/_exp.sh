b() { python bench.py --no-cpu-baseline --steps 20 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  %.1f Mpairs/s' % (d['ms_per_step'], d['value']/1e6), d['config'].get('windows'), d.get('stages_ms'))"; }
for w in 8 4 2; do d=2; [ $w = 8 ] && d=4; python tools/bench_shard.py --logn 20 --world $w --depth $d --steps 40 | cut -c1-70; done
echo "== n=2^20 default"; b --logn 20
echo "== n=2^20 default no-pipeline"; b --logn 20 --no-pipeline
echo "== n=2^18"; b --logn 18
echo "== g2"; b --logn 20 --group 2
echo "== bn254 2^22"; b --logn 22 --curve bn128
