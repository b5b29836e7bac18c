#!/usr/bin/env python3
"""One-off larger parity check (not part of the test suite: the oracle needs tens of seconds): GPU == oracle at n = 2^18 for
both curves and groups, random scalars < 2^256 and generated points; and GLV schedule == plain schedule at n = 2^22."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import numpy as np
import msm_amd
from oracle import oracle as orc
orc.build()
eng = msm_amd.Engine(0)
rng = np.random.default_rng(2026)
for curve in ("bls12381", "bn128"):
    for g in (1, 2):
        n = 1 << (18 if g == 1 else 16)
        C = eng.curve(curve); O = orc.Oracle(curve, g)
        pts = C.gen_points(g, 99 + g, n)
        host = pts.download(n * C.affine_bytes(g))
        sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
        t = time.time(); got = C._multiexp(g, 0, host, sc, 32, n); tg = time.time() - t
        t = time.time(); exp = O.normalize(O.multiexp_affine(host, sc, 32, n)); to = time.time() - t
        print(curve, "g%d" % g, "n=2^%d" % (n.bit_length() - 1), "match" if got == exp else "MISMATCH", "gpu %.3fs oracle %.1fs" % (tg, to), flush=True)
        pts.free()
C = eng.curve("bls12381")
n = 1 << 22
pts = C.gen_points(1, 7, n)
sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8).tobytes()
ds = eng.alloc(n * 32).upload(sc)
eng.set_option("glv", 2); a = C._multiexp(1, 0, pts, ds, 32, n)
eng.set_option("glv", 0); b = C._multiexp(1, 0, pts, ds, 32, n)
print("bls12381 g1 n=2^22 GLV schedule == plain schedule:", a == b)
