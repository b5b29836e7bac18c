#!/usr/bin/env python3
"""Scalars as a prover's witness has them -- many zeros and ones, many small values, the rest uniform -- against uniform scalars:
ms per call (two in flight), entries scattered, and the result cross-checked between the general path and a prepared table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
if os.environ.get("HOT") is not None:
    eng.set_option("hot", int(os.environ["HOT"]))
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
pb = C.prepare_bases(1, pts, n)
g = torch.Generator(device="cuda"); g.manual_seed(11)
uni = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); uni[:, 31] &= 0x3F
def mix(pz, p1, psmall, small_bytes=2, p64=0.0):
    s = uni.clone()
    r = torch.rand(n, device="cuda", generator=g)
    zero = r < pz
    one = (r >= pz) & (r < pz + p1)
    small = (r >= pz + p1) & (r < pz + p1 + psmall)
    mid = (r >= pz + p1 + psmall) & (r < pz + p1 + psmall + p64)
    s[zero] = 0
    s[one] = 0; s[one, 0] = 1
    s[small, small_bytes:] = 0
    s[mid, 8:] = 0
    return s.contiguous()
cases = [("uniform below 2^254", uni), ("50 % zero, 50 % uniform", mix(0.5, 0, 0)), ("30 % zero, 30 % one, 20 % below 2^16, 20 % uniform", mix(0.3, 0.3, 0.2)),
         ("10 % zero, 60 % below 2^16, 30 % uniform", mix(0.1, 0.0, 0.6)), ("all below 2^64", mix(0, 0, 0, p64=1.0)), ("all below 2^128 (batching challenges)", uni.clone().index_fill_(1, torch.arange(16, 32, device="cuda"), 0).contiguous()), ("90 % one, 10 % uniform", mix(0.0, 0.9, 0.0))]
torch.cuda.synchronize()
def timed(fn, steps=40, depth=2):
    fl = []
    def run(k):
        for _ in range(k):
            fl.append(fn())
            if len(fl) == depth: fl.pop(0).wait()
        while fl: fl.pop(0).wait()
    run(20)        # (lets the skew hint settle)
    t0 = time.perf_counter(); run(steps); return (time.perf_counter() - t0) / steps * 1e3
for name, sc in cases:
    ref = C._multiexp(1, 0, pts, sc, 32, n)
    entries = eng.geometry()["entries"]
    ok = pb.multiexp(sc) == ref
    t_gen = timed(lambda: C.multiexp_async(1, pts, sc, 32, n))
    t_one = timed(lambda: C.multiexp_async(1, pts, sc, 32, n), steps=20, depth=1)
    t_prep = timed(lambda: pb.multiexp_async(sc))
    C._multiexp(1, 0, pts, sc, 32, n); C._multiexp(1, 0, pts, sc, 32, n)
    st = {k: round(v, 3) for k, v in eng.timings().items()}
    print(f"{name:52s} entries {entries/1e6:6.2f} M   general {t_gen:6.3f} ms (one at a time {t_one:6.3f})   prepared {t_prep:6.3f} ms   prepared == general: {ok}\n      stages of one general call: {st}", flush=True)
pb.release(); pts.free(); eng.close()
