#!/usr/bin/env python3
"""Where a pipelined small call's time goes on the HOST side: seconds inside the submit call, inside wait (GPU wait + host combine)
and the engine's own host-combine figure, per call, for n = 2^16 .. 2^20 with the bench's number of calls in flight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import torch
import msm_amd
MASK = int(os.environ.get("TOP_MASK", "0x7F"), 16)
for logn, depth in ((16, 3), (17, 3), (18, 3), (20, 2)):
    eng = msm_amd.Engine(0)      # a fresh context per size, as bench.py has
    C = eng.curve("bls12381")
    n = 1 << logn
    eng.set_option("slots", depth)
    pts = C.gen_points(1, 0x5151, n)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda"); sc[:, 31] &= MASK
    torch.cuda.synchronize()
    def run(steps, rec=None):
        fl = []
        for _ in range(steps):
            t0 = time.perf_counter(); fl.append(C.multiexp_async(1, pts, sc, 32, n)); t1 = time.perf_counter()
            if rec is not None: rec[0] += t1 - t0
            if len(fl) == depth:
                t0 = time.perf_counter(); fl.pop(0).wait(); t1 = time.perf_counter()
                if rec is not None: rec[1] += t1 - t0; rec[2] += eng.timings()["host_combine"] * 1e-3
        while fl:
            t0 = time.perf_counter(); fl.pop(0).wait(); t1 = time.perf_counter()
            if rec is not None: rec[1] += t1 - t0; rec[2] += eng.timings()["host_combine"] * 1e-3
    run(20)
    rec = [0.0, 0.0, 0.0]
    steps = 300
    t0 = time.perf_counter(); run(steps, rec); dt = time.perf_counter() - t0
    print(f"n=2^{logn} depth {depth}: {dt/steps*1e3:.3f} ms per call = submit {rec[0]/steps*1e3:.3f} + wait {rec[1]/steps*1e3:.3f} (of which host combine {rec[2]/steps*1e3:.3f}) + loop {(dt-rec[0]-rec[1])/steps*1e3:.3f}", flush=True)
    pts.free()
    eng.close()
