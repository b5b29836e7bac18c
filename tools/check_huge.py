#!/usr/bin/env python3
"""One-off: an MSM beyond the per-pass limit (n = 3 * 2^26 = 201,326,592 pairs, device-resident) through the synchronous entry point,
which slices it internally, against the sum of three explicit 2^26 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import numpy as np
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
m = 1 << 26
n = 3 * m
t = time.time(); pts = C.gen_points(1, 11, n); print("generated %d points in %.1f s" % (n, time.time() - t), flush=True)
rng = np.random.default_rng(7)
ds = eng.alloc(n * 32)
for k in range(3):     # upload the scalars slice by slice (2 GiB each)
    sc = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
    sc[:, 31] &= 0x3F
    eng._chk(eng.L.msm_amd_h2d(eng.ctx, ds.ptr + k * m * 32, sc.tobytes(), m * 32))
t = time.time(); whole = C._multiexp(1, 0, pts, ds, 32, n); dt = time.time() - t
print("n = 3*2^26 in one call: %.1f ms (%.3e pairs/s)" % (dt * 1e3, n / dt), flush=True)
import ctypes
parts = b""
for k in range(3):
    out = ctypes.create_string_buffer(144)
    eng._chk(eng.L.msm_amd_multiexp(eng.ctx, 0, 1, msm_amd.FLAG_DEVICE, pts.ptr + k * m * 96, ds.ptr + k * m * 32, 32, m, out))
    parts += out.raw
print("equals the sum of three 2^26 calls:", msm_amd.combine("bls12381", 1, parts) == whole)
