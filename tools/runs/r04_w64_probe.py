import os, sys, time
sys.path.insert(0, "/root/repo/zprize-wasm-msm_amd/python")
import torch, msm_amd
eng = msm_amd.Engine(0)
C = eng.curve("bls12381")
n = 1 << 20
pts = C.gen_points(1, 0x5151, n)
g = torch.Generator(device="cuda"); g.manual_seed(11)
uni = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g); uni[:, 31] &= 0x3F
s64 = uni.clone(); s64[:, 8:] = 0; s64 = s64.contiguous()
s128 = uni.clone(); s128[:, 16:] = 0; s128 = s128.contiguous()
torch.cuda.synchronize()
def timed(fn, steps=40, depth=2):
    fl = []
    def run(k):
        for _ in range(k):
            fl.append(fn())
            if len(fl) == depth: fl.pop(0).wait()
        while fl: fl.pop(0).wait()
    run(20)
    t0 = time.perf_counter(); run(steps); return (time.perf_counter() - t0) / steps * 1e3
pb = None
if len(sys.argv) > 1:
    pb = C.prepare_bases(1, pts, n)
for name, sc in (("64", s64), ("128", s128), ("uni", uni), ("64 again", s64)):
    if pb is not None:
        pb.multiexp(sc)
    t = timed(lambda: C.multiexp_async(1, pts, sc, 32, n))
    C._multiexp(1, 0, pts, sc, 32, n)
    print(name, "%.3f ms" % t, eng.geometry(), {k: round(v, 3) for k, v in eng.timings().items()}, eng.hints(), flush=True)
