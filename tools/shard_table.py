#!/usr/bin/env python3
"""Per-rank cost table of the multi-GPU split, measured on ONE GPU: for n = 2^16 .. 2^26 and 1, 2, 4, 8 ranks, every split
gp point slices x gw window groups (gp * gw = ranks) is timed as the share of its busiest rank run alone -- slice 0 (the longest)
with window group 0 and with window group gw - 1 (the one that owns the partly filled top window), `depth` calls in flight as
bench.py keeps them -- together with the split msm_amd_dist_plan picks.  The 8-GPU scaling run is the driver's; this table is
what one rank of it has to do, and what the split policy (msm_dist_split in csrc/dist.hip) is fitted to.

  python tools/shard_table.py [--logn 16 18 20 22 24 26] [--worlds 1 2 4 8] [--out gpurun_out/r03_shard_per_rank_table.jsonl]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, nargs="+", default=[16, 18, 20, 22, 24, 26])
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--curve", default="bls12381")
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--budget", type=float, default=0.4, help="seconds of timed work per measurement")
    ap.add_argument("--out", default="")
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    import msm_amd
    eng = msm_amd.Engine(0)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    C = eng.curve(a.curve)
    g = a.group
    asz = C.affine_bytes(g)
    nmax = 1 << max(a.logn)
    t0 = time.time()
    pts = C.gen_points(g, 0x5AAD, nmax)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=gen)
    sc[:, 31] &= 0x7F if a.curve == "bls12381" else 0x3F            # 255 / 254 significant bits, as scalars below the group order have
    torch.cuda.synchronize()
    print(f"# inputs for n = 2^{max(a.logn)} generated in {time.time() - t0:.1f} s", flush=True)
    flat = sc.reshape(-1)
    memo = {}

    def share(count, w, gw, depth):
        """ms per call of window group w of gw on `count` points, `depth` calls in flight"""
        key = (count, w, gw, depth)
        if key in memo:
            return memo[key]
        eng.set_option("slots", depth)
        b, s = pts.view(0, count * asz), flat[:count * 32]

        def run(steps):
            fl = []
            for _ in range(steps):
                fl.append(C.multiexp_shard_async(g, b, s, 32, count, w, gw))
                if len(fl) == depth:
                    fl.pop(0).wait()
            while fl:
                fl.pop(0).wait()

        run(max(3, depth))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(4)
        est = (time.perf_counter() - t1) / 4
        steps = max(4, min(400, int(a.budget / max(est, 1e-5))))
        t1 = time.perf_counter()
        run(steps)
        ms = (time.perf_counter() - t1) / steps * 1e3
        memo[key] = (ms, eng.geometry()["c"])
        return memo[key]

    out = open(a.out, "w") if a.out else None
    print(f"# {a.curve} G{g}; ms per call of the busiest rank's share, run alone on one MI355X", flush=True)
    for logn in a.logn:
        n = 1 << logn
        base = None
        for world in a.worlds:
            plan = eng.dist_plan(a.curve, g, 32, n, world, 0)
            rows = []
            for gw in range(1, world + 1):
                if world % gw:
                    continue
                gp = world // gw
                count = (n + gp - 1) // gp
                # two, three and four calls in flight: what pays depends on the size of the share (bench.py keeps four in flight when the
                # share's accumulation is small enough to run on the slots' own streams, two otherwise)
                per_depth = {}
                for depth in (2, 3, 4):
                    ms0, c = share(count, 0, gw, depth)
                    ms1 = share(count, gw - 1, gw, depth)[0] if gw > 1 else ms0
                    per_depth[depth] = max(ms0, ms1)
                depth = min(per_depth, key=per_depth.get)
                rows.append({"gp": gp, "gw": gw, "points_per_rank": count, "c": c, "ms_depth2": per_depth[2], "ms_depth3": per_depth[3], "ms_depth4": per_depth[4],
                             "calls_in_flight": depth, "ms_busiest": per_depth[depth]})
            best = min(rows, key=lambda r: r["ms_busiest"])
            chosen = next(r for r in rows if r["gw"] == plan["gw"])
            if world == 1:
                base = chosen["ms_busiest"]
            rec = {"logn": logn, "ranks": world, "splits": rows, "planned": {"gp": plan["gp"], "gw": plan["gw"]},
                   "planned_ms": chosen["ms_busiest"], "best": {"gp": best["gp"], "gw": best["gw"]}, "best_ms": best["ms_busiest"],
                   "one_gpu_ms": base, "speedup_planned_before_exchange": (base / chosen["ms_busiest"]) if base else None}
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()
            cells = "  ".join(f"{r['gp']}x{r['gw']}(c={r['c']},d{r['calls_in_flight']}): {r['ms_busiest']:.3f}" for r in rows)
            sp = f"{base / chosen['ms_busiest']:.2f}x" if base else "-"
            print(f"n=2^{logn} ranks={world}: {cells}   planned {plan['gp']}x{plan['gw']} = {chosen['ms_busiest']:.3f} ms ({sp} of one GPU)"
                  f"{'' if (best['gp'], best['gw']) == (plan['gp'], plan['gw']) else f'   BEST {best_str(best)}'}", flush=True)
    # the same measurements by (points per rank, window groups): what the split policy's cost table is made of
    print("# per-share cost: points per rank, window groups -> ms at 2 | 3 | 4 calls in flight (window group 0)")
    shares = {}
    for (count, w, gw, depth), (ms, c) in sorted(memo.items()):
        if w == 0:
            shares.setdefault((count, gw), {})[depth] = ms
    for (count, gw), d in sorted(shares.items()):
        print(f"share n_slice={count} gw={gw}: " + " | ".join(f"{d.get(k, float('nan')):.3f}" for k in (2, 3, 4)))
    if out:
        out.write(json.dumps({"shares": [{"points": c_, "gw": g_, "ms_depth2": d.get(2), "ms_depth3": d.get(3), "ms_depth4": d.get(4)} for (c_, g_), d in sorted(shares.items())]}) + "\n")
    pts.free()
    eng.close()


def best_str(b):
    return f"{b['gp']}x{b['gw']} = {b['ms_busiest']:.3f}"


if __name__ == "__main__":
    main()
