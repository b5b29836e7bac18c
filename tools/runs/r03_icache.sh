#!/bin/bash
# instruction-cache counters of the engine's kernels (the G1 accumulation loop is ~40 KB of code, the reduction's two inlined additions
# ~110 KB; the cache is 64 KB per two CUs): one rocprofv3 --pmc pass per group, one call at a time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 tools/pmc_kernel_counters.py gpurun_out/r03_pmc_icache.json "SQC_ICACHE_REQ,SQC_ICACHE_HITS,SQC_ICACHE_MISSES;SQC_ICACHE_MISSES_DUPLICATE,SQ_IFETCH;SQ_WAIT_INST_ANY,SQ_WAVE_CYCLES,SQ_INSTS_VALU,SQ_BUSY_CYCLES" --steps 5 --warmup 2 > gpurun_out/r03_pmc_icache.log 2>&1; echo rc=$?
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_pmc_icache.json"))
for k, v in d["kernels"].items():
    if "accumulate<" in k or "bucket_reduce" in k or "tile_scatter" in k or "fine_small" in k:
        print(k[:64], {a: round(b) for a, b in v.items() if isinstance(b, (int, float))})
PY
tail -5 gpurun_out/r03_pmc_icache.log
