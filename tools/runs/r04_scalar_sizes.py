#!/usr/bin/env python3
"""Declared scalar sizes other than 32 bytes (the reference's scalarSize argument, build_multiexp.js:251-257): ms per MSM at n = 2^18 / 2^20 for
scalar_size = 8, 16, 24, 31, 32 (uniform over the declared width), two or three calls in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    eng.set_option(k, int(v))
C = eng.curve("bls12381")
for logn, depth in ((18, 3), (20, 2)):
    n = 1 << logn
    eng.set_option("slots", depth)
    pts = C.gen_points(1, 0x5151, n)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    for ss in (8, 16, 24, 31, 32):
        sc = torch.randint(0, 256, (n, ss), dtype=torch.uint8, device="cuda", generator=g)
        if ss == 32:
            sc[:, 31] &= 0x3F
        sc = sc.contiguous()
        torch.cuda.synchronize()
        def run(k):
            fl = []
            for _ in range(k):
                fl.append(C.multiexp_async(1, pts, sc, ss, n))
                if len(fl) == depth: fl.pop(0).wait()
            while fl: fl.pop(0).wait()
        run(8)
        t0 = time.perf_counter(); run(40); ms = (time.perf_counter() - t0) / 40 * 1e3
        C._multiexp(1, 0, pts, sc, ss, n)
        print(f"n=2^{logn} scalar_size={ss:2d}: {ms:.3f} ms per MSM  geometry {eng.geometry()}  stages {dict((k, round(v, 3)) for k, v in eng.timings().items())}  hints {eng.hints()}", flush=True)
    pts.free()
eng.close()
# (appended) the per-window entry point (msm_amd_multiexp_chunk: what the reference's external caller fans out to its workers)
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
g = torch.Generator(device="cuda"); g.manual_seed(12)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 31] &= 0x3F
sc = sc.contiguous(); torch.cuda.synchronize()
for start, size in ((0, 16), (32, 16), (240, 16), (0, 13), (17, 20), (100, 8)):
    C.g1m_multiexpAffine_chunk(pts, sc, 32, n, start, size)
    t0 = time.perf_counter()
    for _ in range(10):
        C.g1m_multiexpAffine_chunk(pts, sc, 32, n, start, size)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"chunk(start={start}, size={size}) at n=2^20: {ms:.3f} ms per call  geometry {eng.geometry()}  stages {dict((k, round(v, 3)) for k, v in eng.timings().items())}", flush=True)
pts.free(); eng.close()
