#!/usr/bin/env python3
"""Does the skew hint (set when a call queued over-long buckets for the tail kernels) decay, or can benign calls keep it alive?
One context: phases of pipelined calls; per phase the mean ms per call, the window layout and how many calls saw queued buckets."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
nmax = 1 << 20
pts = C.gen_points(1, 0x5151, nmax)
def scalars(n, mask):
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda"); s[:, 31] &= mask; return s
def phase(name, n, sc, depth, steps):
    eng.set_option("slots", depth)
    fl = []; t0 = time.perf_counter()
    for _ in range(steps):
        fl.append(C.multiexp_async(1, pts, sc, 32, n))
        if len(fl) == depth: fl.pop(0).wait()
    while fl: fl.pop(0).wait()
    dt = time.perf_counter() - t0
    g = eng.geometry()
    print(f"{name:46s} {dt/steps*1e3:7.3f} ms per call   c={g['c']} W={g['windows']}", flush=True)
s255, s254 = scalars(nmax, 0x7F), scalars(nmax, 0x3F)
eq = s255[:1].repeat(1 << 16, 1).contiguous()
torch.cuda.synchronize()
phase("2^20, 255-bit scalars (fresh context)", nmax, s255, 2, 60)
phase("2^20, 254-bit scalars", nmax, s254, 2, 60)
phase("2^20, 255-bit again", nmax, s255, 2, 60)
phase("2^18, 254-bit scalars, 3 in flight", 1 << 18, s254, 3, 60)
phase("2^20, 254-bit right after", nmax, s254, 2, 60)
phase("2^20, 254-bit, 60 more", nmax, s254, 2, 60)
phase("2^20, 255-bit right after", nmax, s255, 2, 60)
phase("2^20, 255-bit, 60 more", nmax, s255, 2, 60)
phase("2^16, all scalars equal (adversarial)", 1 << 16, eq, 2, 10)
phase("2^20, 255-bit right after", nmax, s255, 2, 30)
phase("2^20, 255-bit, 60 more", nmax, s255, 2, 60)
phase("2^20, 254-bit after that", nmax, s254, 2, 60)
phase("2^20, 254-bit, 60 more", nmax, s254, 2, 60)
