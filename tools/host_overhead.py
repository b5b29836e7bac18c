#!/usr/bin/env python3
"""Host time per call against GPU time per call at small n (round 4): is a stream of small MSMs bound by the GPU or by the one host
thread that submits and collects them?  For every n: (a) the wall time of msm_amd_multiexp_async alone (queueing ~25 launches),
(b) the wall time of msm_amd_wait on a call that has already finished on the GPU (event query + host combine + bookkeeping),
(c) ms per call of one thread keeping `depth` calls in flight, (d) the same with TWO host threads on the same context, each keeping
`depth` calls in flight (ctypes releases the GIL inside the library).  If (d) is markedly better than (c), or (a) + (b) is close to
(c), the host is the limit."""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, nargs="+", default=[14, 16, 17, 18, 20])
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--curve", default="bls12381")
    ap.add_argument("--opt", action="append", default=[])
    a = ap.parse_args()
    import torch
    import msm_amd
    eng = msm_amd.Engine(0)
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    C = eng.curve(a.curve)
    nmax = 1 << max(a.logn)
    pts = C.gen_points(1, 0x5AAD, nmax)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)
    sc = torch.randint(0, 256, (nmax, 32), dtype=torch.uint8, device="cuda", generator=gen)
    sc[:, 31] &= 0x7F if a.curve == "bls12381" else 0x3F
    torch.cuda.synchronize()
    flat = sc.reshape(-1)
    asz = C.affine_bytes(1)
    for logn in a.logn:
        n = 1 << logn
        b, s = pts.view(0, n * asz), flat[:n * 32]
        eng.set_option("slots", 2 * a.depth)
        for _ in range(6):
            C.multiexp_async(1, b, s, 32, n).wait()
        # (a), (b)
        t_sub = t_wait = 0.0
        reps = 50
        for _ in range(reps):
            t0 = time.perf_counter()
            t = C.multiexp_async(1, b, s, 32, n)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            time.sleep(0.003)
            t2 = time.perf_counter()
            t.wait()
            t3 = time.perf_counter()
            t_sub += t1 - t0
            t_wait += t3 - t2
        host = eng.timings()["host_combine"]

        def loop(steps, depth, out, idx):
            fl = []
            t0 = time.perf_counter()
            for _ in range(steps):
                fl.append(C.multiexp_async(1, b, s, 32, n))
                if len(fl) == depth:
                    fl.pop(0).wait()
            while fl:
                fl.pop(0).wait()
            out[idx] = time.perf_counter() - t0

        steps = 300
        out = [0.0, 0.0]
        loop(30, a.depth, out, 0)
        loop(steps, a.depth, out, 0)
        one = out[0] / steps * 1e3
        th = [threading.Thread(target=loop, args=(steps, a.depth, out, k)) for k in range(2)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        two = (time.perf_counter() - t0) / (2 * steps) * 1e3
        print(f"n=2^{logn}: submit {t_sub / reps * 1e3:.3f} ms, collect of a finished call {t_wait / reps * 1e3:.3f} ms (host combine inside: {host:.3f} ms); "
              f"one thread, {a.depth} in flight: {one:.3f} ms per call; two threads, {a.depth} each: {two:.3f} ms per call", flush=True)
    pts.free()
    eng.close()


if __name__ == "__main__":
    main()
