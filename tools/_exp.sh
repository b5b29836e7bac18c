run() { echo "N=1 2^20: $(python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")";
  echo "N=1 2^20 lone: $(python bench.py --no-cpu-baseline --steps 30 --no-pipeline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")";
  echo "N=1 2^16: $(python bench.py --no-cpu-baseline --steps 30 --logn 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")";
  echo "N=1 2^12: $(python bench.py --no-cpu-baseline --steps 30 --logn 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms' % d['ms_per_step'])")";
  python tools/bench_shard.py --logn 20 --world 8 --depth 4 --steps 40 | cut -c1-70;
  python tools/bench_shard.py --logn 20 --world 8 --rank 3 --depth 4 --steps 40 | cut -c1-70;
  python tools/bench_shard.py --logn 20 --world 8 --depth 6 --steps 40 | cut -c1-70;
  python tools/bench_shard.py --logn 20 --world 8 --depth 2 --steps 40 | cut -c1-70;
  python tools/bench_shard.py --logn 20 --world 4 --depth 2 --steps 40 | cut -c1-70; 
  python tools/bench_shard.py --logn 20 --world 2 --depth 2 --steps 40 | cut -c1-70;
  python tools/bench_shard.py --logn 24 --world 8 --depth 4 --steps 20 | cut -c1-70;
  python tools/bench_shard.py --logn 24 --world 8 --depth 2 --steps 20 | cut -c1-70; }
run
