#!/usr/bin/env python3
"""One witness-like scalar mix at n = 2^20 (BLS12-381 G1), general path, two in flight, for a kernel profile (round 4):
   r04_witness_profile.py CASE [steps]    CASE: 0 uniform | 1 30 % zero 30 % one 20 % below 2^16 20 % uniform | 2 10 % zero 60 % below 2^16 30 % uniform
                                                | 3 90 % one 10 % uniform | 4 50 % zero 50 % uniform"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
case = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
eng = msm_amd.Engine(0)
DEPTH = 2
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    if k == "depth":
        DEPTH = int(v); eng.set_option("slots", DEPTH)
    else:
        eng.set_option(k, int(v))
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
g = torch.Generator(device="cuda"); g.manual_seed(11)
uni = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); uni[:, 31] &= 0x3F
def mix(pz, p1, psmall):
    s = uni.clone()
    r = torch.rand(n, device="cuda", generator=g)
    zero = r < pz
    one = (r >= pz) & (r < pz + p1)
    small = (r >= pz + p1) & (r < pz + p1 + psmall)
    s[zero] = 0
    s[one] = 0; s[one, 0] = 1
    s[small, 2:] = 0
    return s.contiguous()
sc = [uni, mix(0.3, 0.3, 0.2), mix(0.1, 0.0, 0.6), mix(0.0, 0.9, 0.0), mix(0.5, 0, 0)][case]
torch.cuda.synchronize()
def run(k, depth=None):
    depth = depth or DEPTH
    fl = []
    for _ in range(k):
        fl.append(C.multiexp_async(1, pts, sc, 32, n))
        if len(fl) == depth: fl.pop(0).wait()
    while fl: fl.pop(0).wait()
run(20)
t0 = time.perf_counter(); run(steps); dt = (time.perf_counter() - t0) / steps * 1e3
C._multiexp(1, 0, pts, sc, 32, n); C._multiexp(1, 0, pts, sc, 32, n)
print(f"case {case}: {dt:.3f} ms per call, {DEPTH} in flight; entries {eng.geometry()['entries'] / 1e6:.2f} M; geometry {eng.geometry()}; hints {eng.hints()}; stages of one call {({k: round(v, 3) for k, v in eng.timings().items()})}")
