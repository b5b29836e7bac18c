#!/usr/bin/env python3
"""Completion time of every call of a pipelined run (two in flight) after a short warmup: where the 20-step figure loses to the 100-step one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import numpy as np
import msm_amd
n = 1 << 20
eng = msm_amd.Engine(0)
eng.set_option("slots", 2)
C = eng.curve("bls12381")
pts = C.gen_points(1, 77, n)
rng = np.random.default_rng(5)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); sc[:, 31] &= 0x7F
ds = eng.alloc(n * 32).upload(sc.tobytes())
def run(k, stamps=None):
    fl = []
    for _ in range(k):
        fl.append(C.multiexp_async(1, pts, ds, 32, n))
        if len(fl) == 2:
            fl.pop(0).wait()
            if stamps is not None: stamps.append(time.perf_counter())
    while fl:
        fl.pop(0).wait()
        if stamps is not None: stamps.append(time.perf_counter())
for warm in (5, 50):
    run(warm)
    time.sleep(0.0)
    st = []
    t0 = time.perf_counter()
    run(30, st)
    d = [round((b - a) * 1e3, 2) for a, b in zip([t0] + st[:-1], st)]
    print(f"warmup {warm}: total {1e3 * (st[-1] - t0) / 30:.3f} ms/step; per completion:", d)
    time.sleep(1.0)   # let the GPU idle before the next round
