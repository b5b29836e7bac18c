#!/bin/bash
# round 4: instruction counts of the two forms of the bucket reduction at n = 2^20, one call at a time (k_bucket_planes + k_planes_combine
# against k_bucket_reduce): SQ_INSTS_VALU and SQ_WAVE_CYCLES / SQ_BUSY_CYCLES per launch, separate --pmc passes (tools/pmc_kernel_counters.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/pmc_kernel_counters.py gpurun_out/r04_pmc_reduce_planes.json "SQ_INSTS_VALU,SQ_WAVES;SQ_WAVE_CYCLES,SQ_BUSY_CYCLES" --steps 5 --warmup 2 > /dev/null 2>&1; echo "planes rc=$?"
python3 tools/pmc_kernel_counters.py gpurun_out/r04_pmc_reduce_old.json "SQ_INSTS_VALU,SQ_WAVES;SQ_WAVE_CYCLES,SQ_BUSY_CYCLES" --steps 5 --warmup 2 --opt reduce_planes=0 > /dev/null 2>&1; echo "old rc=$?"
python3 - <<PY
import json
for f in ("gpurun_out/r04_pmc_reduce_planes.json", "gpurun_out/r04_pmc_reduce_old.json"):
    d = json.load(open(f))
    print(f)
    for k, v in d["kernels"].items():
        if "planes" in k or "bucket_reduce" in k or k.startswith("k_accumulate<"):
            print("  ", k[:70], {a: round(b, 1) for a, b in v.items()})
PY
