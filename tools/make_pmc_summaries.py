#!/usr/bin/env python3
"""profiles/pmc_scatter_<key>.json and profiles/pmc_accumulate_<key>.json -- the figures bench.py quotes as `traffic` / `per_kernel` for
the workload <key> = <curve>_g<group>_2p<logn> -- derived from the per-kernel counter files of tools/pmc_per_kernel.py (and, optionally,
tools/pmc_kernel_counters.py) taken on THAT workload.  The headline's pair is also written under the old names (pmc_scatter.json,
pmc_accumulate.json).

    python3 tools/make_pmc_summaries.py KEY N ENTRIES C WINDOWS fetch_write.json [sq.json]
e.g. python3 tools/make_pmc_summaries.py bls12381_g1_2p20 1048576 17004555 16 17 profiles/r04_pmc_fetch_write_per_kernel_2p20.json
(N pairs, ENTRIES scattered (pair, window) entries, C window bits and WINDOWS as bench.py's JSON line of the same workload reports them)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SORT = ("k_tile_hist", "k_bin_scan", "k_tile_scatter", "k_fine_small", "k_order_hist", "k_order_scatter", "k_big_", "k_items_plan")
POINT_BYTES = {"bls12381_g1": 96, "bn128_g1": 64, "bls12381_g2": 192, "bn128_g2": 128}


def main():
    key, n, entries, c, windows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    fw = json.load(open(sys.argv[6]))["kernels"]
    sq = json.load(open(sys.argv[7]))["kernels"] if len(sys.argv) > 7 else {}
    short = lambda k: k.replace("void msm::", "").replace("msm::", "").split("(")[0]
    per, total, us = {}, 0.0, 0.0
    for k, v in fw.items():
        name = short(k)
        if not name.startswith(SORT):
            continue
        per[name] = {"us": v.get("avg_us"), "moved_MB": v["hbm_MB_fetch_x2"], "GBps_on_moved_bytes": v.get("GBps_on_moved_bytes_fetch_x2"),
                     "frac_of_8TBps": round(v.get("GBps_on_moved_bytes_fetch_x2", 0.0) / 8000.0, 3)}
        total += v["hbm_MB_fetch_x2"]
        us += v.get("avg_us") or 0.0
    alg = n * 32 + entries * 4      # 32 B per scalar + 4 B per scattered (pair, window) entry (bench.py: geometry entries)
    doc = {"workload": f"{key}: n = {n}, {c}-bit windows, {windows} windows (bench.py --no-pipeline), round-4 sort", "window_bits": c, "windows": windows,
           "kernels": " + ".join(sorted(per)),
           "hbm_bytes_per_call": int(round(total * 1e6)), "kernel_us_per_call": round(us, 1),
           "per_kernel": per,
           "algorithmic_bytes_per_call": alg,
           "moved_over_algorithmic": round(total * 1e6 / alg, 2),
           "note": "2 x FETCH_SIZE + WRITE_SIZE per kernel (FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes for 16-byte streaming reads), launch durations "
                   "of the same rocprofv3 passes (the counters slow the kernels by a few per cent against the unprofiled stage figures of bench.py).  The scalars are read "
                   "twice (digits recomputed), the 4-byte entries written once and read once, the index lists written once.",
           "source": os.path.basename(sys.argv[6]) + " (tools/pmc_per_kernel.py: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)"}
    names = [f"pmc_scatter_{key}.json"] + (["pmc_scatter.json"] if key == "bls12381_g1_2p20" else [])
    for nm in names:
        json.dump(doc, open(os.path.join(ROOT, "profiles", nm), "w"), indent=1)
    acc_k, acc = max(((k, v) for k, v in fw.items() if short(k).startswith("k_accumulate<")), key=lambda kv: kv[1].get("avg_us") or 0)
    f, w = acc["FETCH_SIZE_KB_raw"], acc["WRITE_SIZE_KB"]
    point = POINT_BYTES["_".join(key.split("_")[:2])]
    out = {"workload": doc["workload"], "window_bits": c, "windows": windows, "kernel": short(acc_k), "launch_us": acc.get("avg_us"),
           "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
           "hbm_bytes_per_launch": (1.6 * f + w) * 1024, "hbm_bytes_per_launch_if_x2": (2 * f + w) * 1024, "hbm_bytes_per_launch_if_x1": (f + w) * 1024,
           "algorithmic_bytes_per_launch": entries * (point + 4),
           "factor_note": "1.6 x FETCH_SIZE + WRITE_SIZE: the gather of whole points is served at 1.58-1.64 true bytes per reported byte (tools/microbench/gather_fetch.hip, "
                          "profiles/r02_gather_fetch_calibration.txt), not the x2 of a 16-byte stream",
           "source": os.path.basename(sys.argv[6]) + " (tools/pmc_per_kernel.py: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}
    s = next((v for k, v in sq.items() if short(k).startswith("k_accumulate<")), None)
    if s:
        out.update({"SQ_INSTS_VALU_per_launch": s.get("SQ_INSTS_VALU"), "effective_clock_GHz": round(s.get("effective_clock_GHz", 0.0), 3),
                    "SQ_WAIT_ANY": s.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": s.get("SQ_WAIT_INST_ANY"), "SQ_ACTIVE_INST_ANY": s.get("SQ_ACTIVE_INST_ANY"), "SQ_WAVE_CYCLES": s.get("SQ_WAVE_CYCLES"),
                    "sq_note": "SQ counters: " + os.path.basename(sys.argv[7]) + " (tools/pmc_kernel_counters.py, one pass per counter group)"})
    names = [f"pmc_accumulate_{key}.json"] + (["pmc_accumulate.json"] if key == "bls12381_g1_2p20" else [])
    for nm in names:
        json.dump(out, open(os.path.join(ROOT, "profiles", nm), "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("hbm_bytes_per_call", "kernel_us_per_call", "moved_over_algorithmic")}), out["hbm_bytes_per_launch"], out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"])


if __name__ == "__main__":
    main()
