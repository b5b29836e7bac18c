#!/usr/bin/env python3
"""Per-rank cost of a window-sharded MSM, measured on ONE GPU: runs the shard `rank` of `world` alone, `depth` calls in
flight, and prints ms per call.  (The 8-GPU scaling run is the driver's; this shows what one rank of it has to do.)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))
import numpy as np  # noqa: E402

import msm_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--bits", type=int, default=255, help="significant bits of the random scalars (256: half of them carry into a 17th 16-bit window, one bucket holding n/2 entries)")
    ap.add_argument("--same", action="store_true", help="adversarial: every scalar equal (one bucket per window holds all n entries)")
    ap.add_argument("--bits254", action="store_true", help="254-bit scalars: the top 16-bit window then holds 14 bits, its buckets are twice as dense")
    ap.add_argument("--dummy-streams", type=int, default=0,
                    help="create this many idle HIP streams before the engine (shifts the stream-to-hardware-queue mapping: DESIGN.md section 7)")
    ap.add_argument("--torch-first", action="store_true", help="initialise torch's GPU context (null stream) before the engine, as bench.py does")
    ap.add_argument("--sharing", action="store_true", help="print which of the engine's streams share a hardware queue (msm_amd_debug_stream_sharing)")
    ap.add_argument("--rccl", choices=["none", "before", "after"], default="none",
                    help="initialise a one-rank RCCL process group (and run one all_gather) before / after the engine is created: "
                         "the streams RCCL creates shift the stream-to-hardware-queue mapping of whatever is created after them")
    a = ap.parse_args()
    n = 1 << a.logn

    def rccl():
        import socket
        import torch
        import torch.distributed as dist
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        x = torch.zeros(144, dtype=torch.uint8, device=dev)
        y = torch.empty(144, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(y, x)
        torch.cuda.synchronize()

    if a.rccl != "none" or a.torch_first:
        import torch   # torch's HIP runtime must be up before the engine library loads its own (otherwise torch finds no GPU)
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
    if a.rccl == "before":
        rccl()
    if a.dummy_streams:
        import ctypes
        msm_amd.load_library()                      # brings the HIP runtime in; the handle below is the same instance
        hip = ctypes.CDLL("libamdhip64.so")
        keep = []
        for _ in range(a.dummy_streams):
            h = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0   # hipStreamNonBlocking
            keep.append(h)
    eng = msm_amd.Engine(0)
    if a.rccl == "after":
        rccl()
    eng.set_option("slots", a.depth)
    if a.sharing:
        import ctypes
        m = (ctypes.c_uint8 * 121)()
        eng.L.msm_amd_debug_stream_sharing.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        assert eng.L.msm_amd_debug_stream_sharing(eng.ctx, m) == 0
        names = ["A", "S", "B0", "C0", "B1", "C1", "B2", "C2", "B3", "C3", "null"]
        groups, seen = [], set()
        for i in range(11):
            if i in seen:
                continue
            g = [j for j in range(11) if m[i * 11 + j]]
            seen.update(g)
            groups.append("+".join(names[j] for j in g))
        print("streams per hardware queue:", "  ".join(groups))
    for kv in a.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    C = eng.curve("bls12381")
    pts = C.gen_points(1, 77, n)
    rng = np.random.default_rng(5)
    sc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    bits = 254 if a.bits254 else a.bits               # 255 significant bits (BLS12-381's order); --bits254: one bit fewer (BN254's)
    if bits < 256:
        sc[:, bits // 8 + 1:] = 0
        sc[:, bits // 8] &= (1 << (bits % 8)) - 1
    if a.same:
        sc[:] = sc[0]
    ds = eng.alloc(n * 32).upload(sc.tobytes())
    for _ in range(3):
        C.multiexp_shard(1, pts, ds, 32, n, a.rank, a.world)
    t0 = time.perf_counter()
    fl = []
    t_submit = t_wait = 0.0
    for _ in range(a.steps):
        t1 = time.perf_counter()
        fl.append(C.multiexp_shard_async(1, pts, ds, 32, n, a.rank, a.world))
        t2 = time.perf_counter()
        t_submit += t2 - t1
        if len(fl) == a.depth:
            fl.pop(0).wait()
            t_wait += time.perf_counter() - t2
    while fl:
        t2 = time.perf_counter()
        fl.pop(0).wait()
        t_wait += time.perf_counter() - t2
    dt = (time.perf_counter() - t0) / a.steps
    print(f"n=2^{a.logn} shard {a.rank}/{a.world} depth {a.depth}: {dt * 1e3:.3f} ms per call (host: submit {t_submit / a.steps * 1e3:.3f} ms, wait {t_wait / a.steps * 1e3:.3f} ms); "
          f"timings of last call {eng.timings()}")


if __name__ == "__main__":
    main()
