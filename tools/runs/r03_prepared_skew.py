#!/usr/bin/env python3
"""prepared tables under adversarial scalars: every scalar equal (ONE bucket holds all 13 n entries of the merged bucket set), two
distinct scalars, scalars with one hot window -- result against the general path on the same inputs, and the time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
for logn in (16, 18, 20):
    n = 1 << logn
    pts = C.gen_points(1, 0xABCD, n)
    pb = C.prepare_bases(1, pts, n)
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    rnd = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen); rnd[:, 31] &= 0x3F
    one = rnd[:1].repeat(n, 1).contiguous()
    two = rnd[:2].repeat(n // 2, 1).contiguous()
    hot = rnd.clone(); hot[:, 4:7] = 0x5A       # bits 32..55 equal in every scalar: one or two windows with a single hot bucket
    for name, sc in (("random", rnd), ("all equal", one), ("two values", two), ("hot window", hot)):
        torch.cuda.synchronize()
        ref = C._multiexp(1, 0, pts, sc, 32, n)
        for rep in range(3):
            t0 = time.perf_counter(); got = pb.multiexp(sc); dt = time.perf_counter() - t0
        t0 = time.perf_counter(); ref2 = C._multiexp(1, 0, pts, sc, 32, n); dg = time.perf_counter() - t0
        print(f"n=2^{logn} {name:11s} prepared {dt*1e3:8.2f} ms   general {dg*1e3:8.2f} ms   equal: {got == ref == ref2}", flush=True)
        assert got == ref
    pb.release(); pts.free()
eng.close()
