#!/usr/bin/env python3
"""the largest single asynchronous call the C ABI accepts (n = 2^27) against the sum of its two halves; one more point is refused"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 27
t0 = time.time(); pts = C.gen_points(1, 0x7777, n); print(f"points generated in {time.time() - t0:.1f} s", flush=True)
gen = torch.Generator(device="cuda"); gen.manual_seed(27)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen); sc[:, 31] &= 0x3F
torch.cuda.synchronize()
t0 = time.perf_counter(); whole = C.multiexp_async(1, pts, sc, 32, n).wait(); dt = time.perf_counter() - t0
print(f"n = 2^27 in one asynchronous call: {dt*1e3:.1f} ms ({n/dt:.3e} pairs/s), geometry {eng.geometry()}", flush=True)
h = n // 2
halves = b"".join(C._multiexp(1, 0, pts.view(k * h * 96, h * 96), sc[k * h:(k + 1) * h], 32, h) for k in range(2))
print("equals the sum of its halves:", msm_amd.combine("bls12381", 1, halves) == whole, flush=True)
assert msm_amd.combine("bls12381", 1, halves) == whole
pts.free(); eng.close()
