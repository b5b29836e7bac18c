#!/usr/bin/env python3
"""does a context that has once run a small call (which creates the second sort stream, of another priority) stay slower for
full-size calls?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda"); sc[:, 31] &= 0x7F
torch.cuda.synchronize()
def phase(name, m, depth, steps):
    eng.set_option("slots", depth)
    fl = []
    def run(k):
        for _ in range(k):
            fl.append(C.multiexp_async(1, pts, sc, 32, m))
            if len(fl) == depth: fl.pop(0).wait()
        while fl: fl.pop(0).wait()
    run(6); t0 = time.perf_counter(); run(steps); dt = time.perf_counter() - t0
    print(f"{name:52s} {dt/steps*1e3:7.3f} ms per call", flush=True)
for rep in range(2):
    phase("2^20 x2 in flight", n, 2, 100)
phase("2^16 x3 in flight (creates the second sort stream)", 1 << 16, 3, 100)
for rep in range(2):
    phase("2^20 x2 in flight, after the small calls", n, 2, 100)
phase("2^16 x3 in flight again", 1 << 16, 3, 100)
phase("2^20 x2 in flight", n, 2, 100)
