#!/usr/bin/env python3
"""hot-bucket path against the ordinary segment path (option hot = 0) on random witness-like mixes at sizes the oracle cannot reach:
both curves, G1 and G2, n = 2^16 .. 2^20, pipelined and not; every pair of results must be equal."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
rnd = random.Random(2026)
g = torch.Generator(device="cuda"); g.manual_seed(77)
bad = 0
for it in range(24):
    curve = rnd.choice(["bls12381", "bn128"]); group = 1 if rnd.random() < 0.7 else 2
    logn = rnd.choice([16, 17, 18, 19, 20] if group == 1 else [16, 17, 18])
    n = (1 << logn) + rnd.randrange(0, 5000)
    eng = msm_amd.Engine(0)
    C = eng.curve(curve)
    pts = C.gen_points(group, 0x1000 + it, n)
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); s[:, 31] &= 0x1F
    r = torch.rand(n, device="cuda", generator=g)
    edges = sorted(rnd.random() for _ in range(4))
    vals = [rnd.choice([1, 1, 2, 3, 0xFFFF, 1 << 16, (1 << 64) - 1]) for _ in range(2)]
    m0 = r < edges[0] * 0.6
    m1 = (r >= edges[0] * 0.6) & (r < edges[1] * 0.8)
    m2 = (r >= edges[1] * 0.8) & (r < edges[2] * 0.9)
    for mask, v in ((m0, vals[0]), (m1, vals[1])):
        s[mask] = 0
        for k in range(8):
            s[mask, k] = (v >> (8 * k)) & 0xFF
    s[m2, rnd.choice([1, 2, 8]):] = 0
    s = s.contiguous(); torch.cuda.synchronize()
    res = {}
    for hot in (1, 0):
        eng.set_option("hot", hot)
        a = C._multiexp(group, 0, pts, s, 32, n)
        t1, t2 = C.multiexp_async(group, pts, s, 32, n), C.multiexp_async(group, pts, s, 32, n)
        b, c = t1.wait(), t2.wait()
        res[hot] = (a, b, c)
    ok = len({x for v in res.values() for x in v}) == 1
    bad += not ok
    print(f"{it:2d} {curve} G{group} n={n} values {vals} shares {[round(e, 2) for e in edges[:3]]}: {'ok' if ok else 'MISMATCH'}", flush=True)
    pts.free(); eng.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
