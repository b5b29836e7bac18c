#!/usr/bin/env python3
"""128-bit scalars (batching challenges) on a table prepared for 128 bits against a table prepared for the group order's 255 and the general path"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
g = torch.Generator(device="cuda"); g.manual_seed(3)
sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); sc[:, 16:] = 0
sc = sc.contiguous(); torch.cuda.synchronize()
def timed(fn, steps=40, depth=2):
    fl = []
    def run(k):
        for _ in range(k):
            fl.append(fn())
            if len(fl) == depth: fl.pop(0).wait()
        while fl: fl.pop(0).wait()
    run(20); t0 = time.perf_counter(); run(steps); return (time.perf_counter() - t0) / steps * 1e3
ref = C._multiexp(1, 0, pts, sc, 32, n)
print(f"general path                      {timed(lambda: C.multiexp_async(1, pts, sc, 32, n)):6.3f} ms per MSM (two in flight)")
for bits in (255, 128):
    pb = C.prepare_bases(1, pts, n, scalar_bits=bits)
    ok = pb.multiexp(sc) == ref
    print(f"table prepared for {bits:3d}-bit scalars {timed(lambda: pb.multiexp_async(sc)):6.3f} ms   {pb.info()}   equal to the general path: {ok}", flush=True)
    pb.release()
