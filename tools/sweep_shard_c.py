#!/usr/bin/env python3
"""Window size of a SHARDED call's share, measured on one GPU (round 4): for shares of `count` points owning window group w of gw,
ms per call at window_bits = 0 (the engine's own choice) and at every forced c of --c, with 2, 3 and 4 calls in flight -- the busier
of window group 0 and gw - 1.  What pick_window_bits' rule for nshards > 1 (csrc/engine.hip) is fitted to.

  python tools/sweep_shard_c.py --logcount 17 18 19 20 --gw 2 4 8 --c 12 13 14 15 16 [--curve bn128] [--group 2]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logcount", type=int, nargs="+", default=[17, 18, 19, 20])
    ap.add_argument("--gw", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--c", type=int, nargs="+", default=[12, 13, 14, 15, 16])
    ap.add_argument("--depths", type=int, nargs="+", default=[2, 3, 4])
    ap.add_argument("--curve", default="bls12381")
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--budget", type=float, default=0.25)
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
    nmax = 1 << max(a.logcount)
    pts = C.gen_points(g, 0x5AAD, nmax)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=gen)
    sc[:, 31] &= 0x7F if a.curve == "bls12381" else 0x3F
    torch.cuda.synchronize()
    flat = sc.reshape(-1)
    out = open(a.out, "w") if a.out else None

    def share(count, w, gw, depth):
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
        return (time.perf_counter() - t1) / steps * 1e3

    print(f"# {a.curve} G{g}: ms per call of a share (busier of window group 0 and gw - 1), best of {a.depths} calls in flight", flush=True)
    for lc in a.logcount:
        count = 1 << lc
        for gw in a.gw:
            cells = []
            for c in [0] + a.c:
                eng.set_window_bits(c)
                best = None
                for depth in a.depths:
                    ms = max(share(count, 0, gw, depth), share(count, gw - 1, gw, depth) if gw > 1 else 0.0)
                    if best is None or ms < best[0]:
                        best = (ms, depth)
                geom = eng.geometry()
                cells.append({"c_forced": c, "c": geom["c"], "windows": geom["windows"], "ms": best[0], "depth": best[1]})
            rec = {"curve": a.curve, "group": g, "logcount": lc, "gw": gw, "cells": cells}
            if out:
                out.write(json.dumps(rec) + "\n")
                out.flush()
            bestc = min(cells[1:], key=lambda x: x["ms"])
            print(f"count=2^{lc} gw={gw}: " + "  ".join(f"c={x['c_forced'] or 'auto(' + str(x['c']) + ')'}: {x['ms']:.3f} (d{x['depth']})" for x in cells) +
                  f"   best forced c = {bestc['c_forced']} ({bestc['ms']:.3f}; auto {cells[0]['ms']:.3f})", flush=True)
    eng.set_window_bits(0)
    pts.free()
    eng.close()


if __name__ == "__main__":
    main()
