#!/usr/bin/env python3
"""bench.py -- BLS12-381 G1 MSM throughput (scalar-point pairs/s) at n = 2^20 on N MI355X.

A "step" is one MSM over the same n = 2^20 synthetic (scalar, point) pairs, inputs resident in
HBM before the timed region.  N = 1: one call of the engine.  N > 1 (one process per GPU,
torch.distributed over RCCL): the Pippenger windows are sharded over the ranks, each rank returns
one Jacobian partial, an all_gather over xGMI (144 B per rank) exchanges them and the host adds
them -- strong scaling of the same MSM.

Prints ONE JSON line on rank 0 (contract in the task statement), with a `roofline` object for the
dominant kernel (bucket accumulation) and a `cpu_baseline` object (the C oracle = a port of the
reference algorithm, single thread, same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

R_BLS = 52435875175126190479447740508185965837690552500527637822603658699938581184513
R_BN = 21888242871839275222246405745257275088548364400416034343698204186575808495617
R_ORDER = {"bls12381": R_BLS, "bn128": R_BN}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TMADS = 36.3         # v_mad_u64_u32 per second, measured at 2.4 GHz (profiles/r01_valu_rates_microbench.txt)
# v_mad_u64_u32 per XYZZ mixed addition in the bucket kernels (6 products + 2 squarings + one fused two-product reduction):
# NL-limb product = 2 NL^2, squaring = NL(NL+1)/2 + NL^2, fused = 3 NL^2; BLS12-381: NL = 14, BN254: NL = 9 (field_u.hpp)
# G2 (an Fp2 element on a lane pair, field_u.hpp Fp2d): per LANE a product is one double-width accumulation (2 NL^2 + NL^2 for its
# reduction), a square one single-width product (2 NL^2; BN254's tight radix squares by the product), the fused form four half
# products and one reduction (5 NL^2); an addition occupies two lanes.
MADS_PER_ADD = {("bls12381", 1): 6 * 392 + 2 * 301 + 588, ("bn128", 1): 6 * 162 + 2 * 126 + 243,
                ("bls12381", 2): 2 * (6 * 588 + 2 * 392 + 980), ("bn128", 2): 2 * (6 * 243 + 2 * 243 + 405)}
VALU_PEAK_IDEAL_TMADS = 39.3   # the same chip at an ideal 4.0 cycles per wave-instruction: 1024 SIMDs x 16 lanes x 2.4 GHz


def uniform_scalars_mod_r(n, seed, r=R_BLS):
    """n scalars uniform in [0, r), 32 bytes little-endian each (rejection sampling on the bit length of r)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4), dtype=np.uint64)
    r_limbs = [(r >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]
    todo = np.arange(n)
    while todo.size:
        cand = rng.integers(0, 1 << 64, size=(todo.size, 4), dtype=np.uint64)
        bits = (r - 1).bit_length()
        for i in range(4):
            lb = min(64, max(0, bits - 64 * i))
            cand[:, i] &= np.uint64((1 << lb) - 1) if lb < 64 else np.uint64((1 << 64) - 1)
        # lexicographic compare cand < r from the top limb
        lt = np.zeros(todo.size, dtype=bool)
        eq = np.ones(todo.size, dtype=bool)
        for i in (3, 2, 1, 0):
            lt |= eq & (cand[:, i] < np.uint64(r_limbs[i]))
            eq &= cand[:, i] == np.uint64(r_limbs[i])
        out[todo[lt]] = cand[lt]
        todo = todo[~lt]
    return out.view(np.uint8).reshape(n, 32)


def count_gpus_sysfs():
    """GPUs of this node as the kernel driver lists them: the KFD topology nodes with SIMDs (CPU nodes have simd_count 0).  Reads sysfs
    only -- no HIP, no torch: the parent of `--gpus N` must not initialise a GPU runtime before it starts its ranks.  None when the
    topology is not readable (no amdgpu driver in this container, a different kernel layout): the caller then skips its pre-check."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    count = 0
    seen = False
    for nd in nodes:
        try:
            with open(os.path.join(base, nd, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        if int(props.get("simd_count", "0")) > 0:
            count += 1
    return count if seen else None


def launch_ranks(n):
    """Parent of a `python bench.py --gpus N` run: N child ranks (one process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    their environment, exactly what torch.distributed.run would set), rank 0's stdout forwarded, exit code = the worst child's.
    The parent never imports torch, the engine or anything else that could start a GPU runtime: the GPUs are counted from sysfs
    (MSM_BENCH_FAKE_GPU_COUNT overrides the count: launcher tests), and where sysfs says nothing the ranks find out themselves --
    a rank whose LOCAL_RANK has no GPU fails, and the first failing child ends the job."""
    import socket
    import subprocess
    backend = os.environ.get("MSM_BENCH_BACKEND", "nccl")
    if backend == "nccl" and os.environ.get("MSM_BENCH_LAUNCH_ONLY", "") != "1":
        fake = os.environ.get("MSM_BENCH_FAKE_GPU_COUNT")
        have = int(fake) if fake else count_gpus_sysfs()
        if have is not None and have < n:
            print(f"bench.py: --gpus {n} asks for {n} ranks over RCCL but this node shows {have} GPU(s); "
                  "(MSM_BENCH_BACKEND=gloo rehearses the N > 1 path with ranks sharing GPUs)", file=sys.stderr)
            assert "torch" not in sys.modules, "the launcher must not import torch"
            return 2
    assert "torch" not in sys.modules and "msm_amd" not in sys.modules, "the launcher must not import torch or the engine"
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # poll all children: the first one to exit non-zero (or the deadline) ends the job -- the others would sit in a collective until
    # the launcher's own timeout (waiting for them in rank order noticed a crash of rank k > 0 only after rank 0 had timed out)
    rc = 0
    deadline = time.time() + float(os.environ.get("MSM_BENCH_LAUNCH_TIMEOUT", "1500"))
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = (bad[0] if bad and bad[0] > 0 else 1)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 timed steps (0.3 s at N = 1) -- the timed region starts with an empty pipeline and ends with a drained one, which
    # costs ~4 ms whatever K is (10 steps: 3.05 ms per step, 30: 2.80, 100: 2.70 on the same box)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--logn", type=int, default=20, help="log2 of the number of pairs (BASELINE config: 20)")
    ap.add_argument("--curve", default="bls12381")
    ap.add_argument("--group", type=int, default=1)
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option (msm_amd_set_option), e.g. reduce_l=32; for experiments")
    ap.add_argument("--host-buffers", action="store_true", help="pass host memory each call (PCIe-inclusive rate; informational, never the headline value)")
    ap.add_argument("--resident-bases", action="store_true", help="with --host-buffers: keep the bases on the GPU, copy only the scalars")
    ap.add_argument("--depth", type=int, default=0, help="calls in flight when pipelined (default 2; 4 for more than four ranks)")
    ap.add_argument("--no-pipeline", action="store_true", help="one call at a time (latency) instead of two in flight (throughput)")
    ap.add_argument("--scalar-bits", type=int, default=0, help="scalars of this many significant bits in their 32 bytes (uniform below 2^bits; default 0: uniform below the "
                    "group order, the metric's workload) -- 128-bit challenges, 64-bit witness values: prints its own line")
    ap.add_argument("--prepared", action="store_true", help="the bases are PREPARED once outside the timed region (msm_amd_prepare_bases: the window table of a prover's "
                    "fixed SRS) and every step is msm_amd_multiexp_prepared; prints its own line -- the default line stays the general path (N = 1 only)")
    args = ap.parse_args()

    # --gpus N > 1 without a launcher around us (python bench.py --gpus N): start the N ranks ourselves.  This happens before torch,
    # the engine library or anything else that touches the GPU is imported: a process that has initialised the GPU must not exec or
    # fork workers.  Under `python -m torch.distributed.run` (the driver's form) WORLD_SIZE is already set and this is skipped.
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to run a mislabelled job")

    # N > 1: RCCL (through torch's stream pools) creates dozens of HIP streams before the engine creates its own, and HIP maps all of
    # them onto four hardware queues by default; the engine's sort / accumulation / reduction streams then share queues they should
    # not (measured on one GPU with a one-rank RCCL group: 1.15 -> 1.42 ms per call for a 4-way shard, 1.57 -> 1.96 for 2-way).  Eight
    # hardware queues restore the figures (profiles/r01_rccl_hw_queues.log).  Must be set before the HIP runtime starts; not for
    # N = 1, where four queues are the faster setting.
    # MSM_BENCH_FORCE_DIST=1 takes the N > 1 code path with a world of one (RCCL group, window shard 0 of 1, all_gather of one
    # partial): the whole multi-GPU flow rehearsed on a one-GPU box
    force_dist = os.environ.get("MSM_BENCH_FORCE_DIST", "") == "1"
    if os.environ.get("MSM_BENCH_LAUNCH_ONLY", "") == "1":     # launcher test (CPU): say who we are and leave before any GPU use
        print(json.dumps({"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "master": os.environ.get("MASTER_ADDR"), "argv_gpus": args.gpus}), flush=True)
        return
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or force_dist:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    sharded = world > 1 or force_dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MSM_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks then share GPUs
    # and the 144-byte partials travel through host memory); the driver's runs use RCCL ("nccl").
    backend = os.environ.get("MSM_BENCH_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    gpu_index = local_rank if backend == "nccl" else local_rank % max(1, ngpu)
    if gpu_index >= ngpu:
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local_rank}) has no GPU: this node shows {ngpu} GPU(s); refusing to run a mislabelled job")
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        torch.cuda.set_device(gpu_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", gpu_index))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    dev = torch.device("cuda", gpu_index)
    torch.cuda.set_device(dev)
    xdev = dev if backend == "nccl" else torch.device("cpu")    # where the partials are exchanged

    n = 1 << args.logn
    eng = msm_amd.Engine(gpu_index)           # fails loudly without the HIP library / a GPU
    if args.window_bits:
        eng.set_window_bits(args.window_bits)
    for kv in args.opt:      # (before the split of the inputs is planned: "dist_gw" changes it)
        name, val = kv.split("=")
        eng.set_option(name, int(val))
    C = eng.curve(args.curve)
    g = args.group
    jsz = C.jacobian_bytes(g)

    # synthetic inputs: P_i = k_i * G (GPU generated), scalars uniform in [0, r).  N = 1: all n of them.  N > 1: every rank generates
    # and holds ONLY ITS SLICE of the points and scalars (gp point slices x gw window groups, msm_amd_dist_plan: 6 GiB of bases at
    # n = 2^26 are 0.75 GiB per GPU at gp = 8) -- the placement a prover with a sharded SRS has; the ranks of one slice (gw > 1) hold
    # the same slice.  MSM_BENCH_GW forces the number of window groups (gw = world: windows only, every rank holds everything).
    if os.environ.get("MSM_BENCH_GW"):
        eng.set_option("dist_gw", int(os.environ["MSM_BENCH_GW"]))
    if args.prepared and sharded:
        eng.set_option("dist_gw", 1)      # prepared tables in a group: the points are cut along the points only, one table per rank
    plan = eng.dist_plan(args.curve, g, 32, n, world, rank) if sharded else {"gp": 1, "gw": 1, "first": 0, "count": n}
    first, count = plan["first"], plan["count"]
    pts = C.gen_points(g, 0xC0FFEE02, count, first)
    sc_host = uniform_scalars_mod_r(n, 0xC0FFEE02, R_ORDER[args.curve] if not args.scalar_bits else (1 << args.scalar_bits))[first:first + count]      # below the curve's own group order
    d_scalars = torch.from_numpy(sc_host).to(dev)
    torch.cuda.synchronize()

    prep = None
    if args.prepared:
        if args.host_buffers:
            sys.exit("bench.py: --prepared is a resident-inputs mode")
        t_prep = time.perf_counter()
        prep = C.prepare_bases(g, pts, count)          # (N > 1: the table of this rank's slice)
        torch.cuda.synchronize()
        t_prep = time.perf_counter() - t_prep
    h_pts = h_sc = None
    if args.host_buffers:
        import numpy as np
        h_pts = np.frombuffer(pts.download(count * C.affine_bytes(g)), dtype=np.uint8).copy()
        h_sc = np.ascontiguousarray(sc_host)

    # N > 1: the exchange of the window partials runs behind the C ABI (msm_amd_dist_*: the engine's own RCCL communicator, ONE
    # ncclAllGather per call) -- the path a Node host takes.  torch.distributed only carries the 128-byte id, the barrier and the MAX
    # over ranks.  The gloo rehearsal (ranks sharing one GPU, which RCCL refuses) and a failed msm_amd_dist_init fall back to the
    # torch.distributed form of the same exchange (msm_amd.dist), and the JSON line says which one ran.
    exchange = None
    if sharded:
        exchange = "torch.distributed all_gather_into_tensor (msm_amd.dist)"
        # (MSM_AMD_RCCL_LIB names another collective library for the engine -- the tests' shared-memory stand-in: then the C-ABI exchange is
        # rehearsed under the gloo backend too, ranks sharing one GPU, which RCCL itself refuses)
        if backend == "nccl" or os.environ.get("MSM_AMD_RCCL_LIB"):
            # the ranks agree on the outcome: rank 0 broadcasts the id or a failure marker unconditionally, and the C-ABI exchange is
            # used only if msm_amd_dist_init succeeded on EVERY rank (a rank falling back on its own would leave the others in
            # ncclCommInitRank or in a collective nobody else enters)
            box = [None]
            if rank == 0:
                try:
                    box = [msm_amd.dist_unique_id()]
                except msm_amd.MsmAmdError as e:
                    print(f"bench.py: msm_amd_dist_unique_id failed ({e})", file=sys.stderr)
            dist.broadcast_object_list(box, src=0)
            ok = 0
            if box[0] is not None:
                try:
                    eng.dist_init(rank, world, box[0])
                    ok = 1
                except msm_amd.MsmAmdError as e:
                    print(f"bench.py: msm_amd_dist_init failed on rank {rank} ({e})", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int32, device=xdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                exchange = "C ABI: msm_amd_multiexp_dist_local_async / msm_amd_dist_wait (ncclAllGather on the engine's own communicator)"
            else:
                if ok:
                    eng.dist_destroy()
                if rank == 0:
                    print("bench.py: exchanging through torch.distributed on every rank", file=sys.stderr)
    cabi = bool(exchange and exchange.startswith("C ABI"))
    rccl_ranks = eng.dist_comm_count()[0] if cabi else (dist.get_world_size() if sharded else None)     # ncclCommCount of the engine's own communicator
    if cabi and args.prepared:
        exchange = "C ABI: msm_amd_multiexp_dist_prepared_async / msm_amd_dist_wait (every rank the prepared table of its slice; ncclAllGather on the engine's own communicator)"
    gw = plan["gw"]

    def step():
        if cabi and prep is not None:
            return prep.multiexp_dist(d_scalars)
        if cabi:
            return C.multiexp_dist_local(g, pts, d_scalars, 32, count, n)
        if args.host_buffers:
            return C._multiexp(g, 0, pts if args.resident_bases else h_pts, h_sc, 32, n)
        if prep is not None:
            return prep.multiexp(d_scalars)
        if not sharded:
            return C._multiexp(g, 0, pts, d_scalars, 32, n)
        # this rank's windows of its slice; ONE all_gather of world x 144 B over RCCL/xGMI; host adds the partials
        return mdist.gather_and_combine(args.curve, g, C.multiexp_shard(g, pts, d_scalars, 32, count, rank % gw, gw), xdev)

    def sync():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    result = None
    stage_acc = {}

    def collect():
        for k, v in eng.timings().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v

    pipelined = not args.no_pipeline and not args.host_buffers
    # calls in flight: 2 hide the latency-bound bucket reduction behind the next accumulation on one GPU; when the
    # windows are sharded 8 ways the accumulation per rank shrinks but the reduction latency does not, so more are needed
    # (tools/bench_shard.py on one GPU: 2 in flight is best up to 4 ranks, 4 at 8 ranks)
    # How many: one probe call tells what a call (of this rank) consists of -- profiles/r03_shard_per_rank_table.txt measured every
    # share with two, three and four in flight.  A call whose accumulation is small -- fewer than 327 680 buckets, so that it runs on
    # its slot's own stream beside the others', and not in the 8-bit-window regime, whose 64-lane slices fill the chip already --
    # is best served by three in flight up to ~10 M (pair, window) entries: n = 2^15 ... 2^18 on one GPU (2^18 at
    # 14-bit windows: 1.06 | 0.92 | 1.01 ms with 2 | 3 | 4 in flight), the window groups of a sharded n <= 2^20.  Everything else: two.
    depth = 1
    if pipelined:
        depth = args.depth or 2
        if not args.depth and not args.host_buffers:
            if prep is not None:
                prep.multiexp(d_scalars)
            elif sharded:
                C.multiexp_shard(g, pts, d_scalars, 32, count, rank % plan["gw"], plan["gw"])
            else:
                C._multiexp(g, 0, pts, d_scalars, 32, n)
            gm = eng.geometry()
            small_acc = gm["windows_owned"] * gm["buckets_per_window"] < 327680 and gm["buckets_per_window"] >= 4096
            if prep is not None:
                # (round 4: n = 2^18 prepared, 3.9 M entries: 1.02 | 0.91 | 1.07 ms with 2 | 3 | 4 in flight; profiles/r04_prepared_small_n.txt)
                want = (4 if gm["entries"] < 3500000 else (3 if gm["entries"] < 10000000 else 2)) if small_acc else 2
            else:
                # (round 4, with the cheaper reduction: three beat four at every small size -- n = 2^15: 0.427 | 0.323 | 0.386 | 0.375 ms with
                # 2 | 3 | 4 | 5 in flight, n = 2^16: 0.527 | 0.418 | 0.483 | 0.459; profiles/r04_depth_small_n_planes.txt)
                want = (3 if gm["entries"] < 10000000 else 2) if small_acc else 2
            if sharded:
                flag = torch.tensor([want], dtype=torch.int32, device=xdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank keeps the same number of calls in flight
                want = int(flag.item())
            depth = want
    eng.set_option("slots", max(depth, 1))

    xch = mdist.PartialExchange(args.curve, g, device=xdev) if sharded else None
    gathering = []   # all_gathers started and not yet collected (at most one is left running behind the loop)

    def finish(ticket):
        out = ticket.wait()      # C ABI form: waits for this rank's partial, all-gathers, adds -- the combined result
        if not sharded or cabi:
            return out
        # this rank's partial is ready: ONE all_gather of world x 144 B over RCCL/xGMI, started asynchronously; the
        # previous call's gathered partials are collected and added on the host meanwhile
        gathering.append(xch.submit(out))
        return xch.result(gathering.pop(0)) if len(gathering) > 1 else None

    def run(steps):
        """`steps` MSMs the way the timed region runs them: every one submitted AND collected before this returns"""
        res = None
        if pipelined:
            # `depth` steps in flight: the bucket reduction of step i (stream B) overlaps the sort + accumulation of step i+1 (stream A)
            inflight = []
            for _ in range(steps):
                if prep is not None and sharded:
                    if not cabi:
                        sys.exit("bench.py: --prepared with N > 1 needs the C-ABI exchange (msm_amd_dist_init failed)")
                    inflight.append(prep.multiexp_dist_async(d_scalars))
                elif prep is not None:
                    inflight.append(prep.multiexp_async(d_scalars))
                elif not sharded:
                    inflight.append(C.multiexp_async(g, pts, d_scalars, 32, n))
                else:
                    inflight.append(C.multiexp_dist_local_async(g, pts, d_scalars, 32, count, n) if cabi else C.multiexp_shard_async(g, pts, d_scalars, 32, count, rank % gw, gw))
                if len(inflight) == depth:
                    res = finish(inflight.pop(0))
                    collect()
            while inflight:
                res = finish(inflight.pop(0))
                collect()
            while gathering:
                res = xch.result(gathering.pop(0))
        else:
            for _ in range(steps):
                res = step()
                collect()
        return res

    # warmup: the same path as the timed region, and at least one call per slot in flight -- a slot allocates its scratch buffers at its
    # first call (13 GB at n = 2^26), which a warmup of synchronous calls left to the timed region for every slot but the first
    warm = run(max(args.warmup, depth if pipelined else 0))
    result = warm if warm is not None else result
    stage_acc.clear()
    sync()
    t0 = time.perf_counter()
    res = run(args.steps)
    result = res if res is not None else result
    sync()
    elapsed = time.perf_counter() - t0
    hints = eng.hints()        # the context's call-to-call state as the timed region left it (wide / skew layouts, work list)
    if sharded:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # two more calls, one at a time and outside the timed region: stage durations without the other call's kernels
    # running beside them (under pipelining the event intervals of a stage include whatever shares the GPU with it)
    iso = {}
    if pipelined and not sharded:
        for _ in range(2):
            step()
            sync()
            iso = dict(eng.timings())
    # latency: individually timed synchronous calls (wall time of one MSM call, median of >= 5 runs)
    lat_ms = []
    if not sharded and not args.host_buffers:
        for _ in range(7):
            sync()
            t1 = time.perf_counter()
            step()
            lat_ms.append((time.perf_counter() - t1) * 1e3)
        lat_ms.sort()
    geom = eng.geometry()
    stages = {k: v / args.steps for k, v in stage_acc.items()}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n * args.steps / elapsed
        aff = C.affine_bytes(g)
        # roofline of the dominant kernel (bucket accumulation): algorithmic bytes per launch =
        # scattered entries x (affine point + 4-byte index)   [SURVEY.md 8(d): B_accum = W*(point+4)]
        alg_bytes = geom["entries"] * (aff + 4)
        acc_ms = stages.get("accumulate", 0.0)
        achieved = alg_bytes / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else 0.0
        # HBM bytes of one k_accumulate launch from the hardware counters (rocprofv3 --pmc, its own run: profiles/pmc_accumulate.json);
        # only quoted for the workload it was measured on
        # only quoted for the workload they were measured on: profiles/pmc_accumulate_<curve>_g<group>_2p<logn>.json (tools/make_pmc_summaries.py;
        # the headline's also under the old name profiles/pmc_accumulate.json), window layout checked
        traffic = None
        key = f"{args.curve}_g{g}_2p{args.logn}"
        plain = not sharded and prep is None

        def counters(kind):
            for name in (f"pmc_{kind}_{key}.json",) + ((f"pmc_{kind}.json",) if key == "bls12381_g1_2p20" else ()):
                path = os.path.join(ROOT, "profiles", name)
                if plain and os.path.exists(path):
                    try:
                        doc = json.load(open(path))
                    except Exception:
                        continue
                    if doc.get("window_bits", geom["c"]) == geom["c"] and doc.get("windows", geom["windows"]) == geom["windows"]:
                        return doc
            return None

        doc_a, doc_s = counters("accumulate"), counters("scatter")
        if doc_a:
            traffic = doc_a.get("hbm_bytes_per_launch")
        # scatter stage (histogram + scan/order + scatter): B_scatter = 32 + 4*W per pair
        sc_ms = stages.get("histogram", 0) + stages.get("scan_order", 0) + stages.get("scatter", 0)
        iso_src = iso if iso else stages
        acc_iso = iso_src.get("accumulate", 0.0)
        sc_iso = iso_src.get("histogram", 0) + iso_src.get("scan_order", 0) + iso_src.get("scatter", 0)
        sc_bytes = n * 32 + geom["entries"] * 4
        sc_traffic = doc_s.get("hbm_bytes_per_call") if doc_s else None
        sc_per_kernel = doc_s.get("per_kernel") if doc_s else None     # {kernel: {us, moved MB, GB/s on moved bytes}}: measured with rocprofv3, this workload only
        # what a TWO-LEVEL partition can reach on this figure of merit: it reads the scalars twice (digits are recomputed, never stored),
        # writes the 4-byte entries, reads them, writes the index lists -- 64 + 12 W bytes per pair against the algorithmic 32 + 4 W --
        # and the chip's streaming-copy rate is ~6.3 of the 8 TB/s
        w_eff = geom["entries"] / max(1, n)
        sc_ceiling = (6300.0 / HBM_PEAK_GBS) * (32 + 4 * w_eff) / (64 + 12 * w_eff)
        mads = MADS_PER_ADD.get((args.curve, g))
        out = {
            "metric": ("BLS12-381 G1 MSM throughput (scalar-point pairs/sec) at n=2^20" if (args.curve, g, args.logn) == ("bls12381", 1, 20)
                       else f"{args.curve} G{g} MSM throughput (scalar-point pairs/sec) at n=2^{args.logn}") + (" -- PREPARED BASES (not the headline path)" if prep is not None else "") + (f" -- {args.scalar_bits}-BIT SCALARS (not the headline workload)" if args.scalar_bits else ""),
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32 limbs (381-bit Montgomery integers)" if args.curve == "bls12381" else "u32 limbs (254-bit Montgomery integers)",
            "data": "synthetic: P_i = k_i*G (GPU-generated), scalars uniform in " + (f"[0,2^{args.scalar_bits}); " if args.scalar_bits else "[0,r); ") + ("inputs in HOST memory, copied over PCIe inside every call" if args.host_buffers else "inputs resident in HBM"),
            "config": {"workload": f"{args.curve} G{g} MSM n=2^{args.logn}, 32-byte scalars", "window_bits": geom["c"],
                       "windows": geom["windows"], "buckets_per_window": geom["buckets_per_window"],
                       "calls_in_flight": depth,
                       "engine_options": args.opt or None,   # --opt NAME=VALUE given on the command line (experiments; none for the judged line)
                       "bases": ("caller-supplied per call (the reference's interface: msm_amd_multiexp)" if prep is None else
                                 {"form": "prepared once outside the timed region (msm_amd_prepare_bases); every step is msm_amd_multiexp_prepared",
                                  **prep.info(), "prepare_s": t_prep}),
                       "parallelism": "1 GPU" if not sharded else f"{plan['gp']} point slices x {plan['gw']} window groups over {world} GPUs (every rank holds only its slice) + ONE RCCL all_gather of the partials",
                       "split": None if not sharded else {"gp": plan["gp"], "gw": plan["gw"], "points_per_rank": count},
                       "exchange": exchange, "rccl_ranks": rccl_ranks,
                       "hints_after_timed_region": hints},
            "mode": (f"pipelined({depth}): {depth} calls in flight, every call collected inside the timed region" if pipelined else "one call at a time"),
            # the dominant kernel is integer-ALU bound: its roofline is the VALU issue rate of v_mad_u64_u32 (MADS_PER_ADD of them per
            # bucket addition, field_u.hpp / curve_u.hpp).  Peak: measured on this chip with tools/microbench/valu_rates.hip
            # (profiles/r01_valu_rates_microbench.txt: 36.3 T mad/s at 2.4 GHz -- the guide lists no integer multiply-add peak; the chip
            # clocks this kernel at ~1.9-2.0 GHz, which puts the practical ceiling of `frac` at ~0.65-0.7).  The HBM fraction on
            # algorithmic bytes (entries x (affine point + 4-byte index)) is kept beside it: small by construction.
            "roofline": ({"kernel": "k_accumulate (bucket accumulation)", "bound": "valu",
                          "achieved": geom["entries"] * mads / (acc_ms * 1e-3) / 1e12, "peak": VALU_PEAK_TMADS, "unit": "T mad/s",
                          "frac": geom["entries"] * mads / (acc_ms * 1e-3) / 1e12 / VALU_PEAK_TMADS,
                          "frac_one_call_at_a_time": (geom["entries"] * mads / (acc_iso * 1e-3) / 1e12 / VALU_PEAK_TMADS) if acc_iso > 0 else None,
                          "mads_per_bucket_addition": mads,
                          "peak_ideal": VALU_PEAK_IDEAL_TMADS, "frac_of_peak_ideal": geom["entries"] * mads / (acc_ms * 1e-3) / 1e12 / VALU_PEAK_IDEAL_TMADS,
                          "peak_note": "peak = measured v_mad_u64_u32 rate (4.34 cycles per wave-instruction at 2.4 GHz); peak_ideal = 4.0 cycles",
                          "peak_source": "measured: tools/microbench/valu_rates.hip, profiles/r01_valu_rates_microbench.txt (v_mad_u64_u32, 2.4 GHz)",
                          "traffic": traffic, "traffic_note": "HBM-side bytes per launch from rocprofv3 --pmc: 1.6 x FETCH_SIZE + WRITE_SIZE -- the guide's x2 on FETCH_SIZE holds for streaming reads; for this kernel's access pattern (random 96-byte records, 16-byte loads) tools/microbench/gather_fetch.hip measures 1.58-1.64 (profiles/r02_gather_fetch_calibration.txt); measured on this workload only",
                          "avg_launch_ms": acc_ms, "avg_launch_ms_one_call_at_a_time": acc_iso,
                          "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": alg_bytes,
                                  "achieved_one_call_at_a_time": (alg_bytes / (acc_iso * 1e-3) / 1e9) if acc_iso > 0 else None}}
                         if mads and acc_ms > 0 else
                         {"kernel": "k_accumulate (bucket accumulation, integer-ALU bound)", "bound": "hbm", "achieved": achieved,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                          "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": acc_ms,
                          "avg_launch_ms_one_call_at_a_time": acc_iso}),
            "roofline_scatter": {"kernels": "k_tile_hist + k_bin_scan[_sums/_ranges] + k_tile_scatter + k_fine_small (k_big_* for spans beyond one workgroup) + k_order_hist + k_order_scatter [+ k_items_plan] (HBM-bound stage, csrc/sort_kernels.hpp)", "bound": "hbm",
                                 "achieved": (sc_bytes / (sc_ms * 1e-3) / 1e9) if sc_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (sc_bytes / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sc_ms > 0 else 0.0,
                                 "algorithmic_bytes": sc_bytes, "avg_ms": sc_ms,
                                 # what the hardware counters say these kernels move (profiles/pmc_scatter.json), this workload only
                                 "traffic": sc_traffic,
                                 "per_kernel": sc_per_kernel,
                                 "avg_ms_one_call_at_a_time": sc_iso,
                                 "frac_one_call_at_a_time": (sc_bytes / (sc_iso * 1e-3) / 1e9 / HBM_PEAK_GBS) if sc_iso > 0 else None,
                                 "traffic_frac_of_peak": (sc_traffic / (sc_iso * 1e-3) / 1e9 / HBM_PEAK_GBS) if (sc_traffic and sc_iso > 0) else None,
                                 "ceiling": {"frac": sc_ceiling, "derivation": "a two-level partition moves at least 64 + 12 W bytes per pair (scalars read twice, 4-byte entries written and read, "
                                             "index lists written) for the algorithmic 32 + 4 W, at ~6.3 TB/s of streaming copies: (6.3 / 8) x (32 + 4 W) / (64 + 12 W), W = entries per pair"}},
            "latency": ({"what": ("wall time of one synchronous msm_amd_multiexp_prepared call" if prep is not None else "wall time of one synchronous msm_amd_multiexp call") + ", inputs resident, nothing else in flight",
                         "runs": len(lat_ms), "median_ms": lat_ms[len(lat_ms) // 2], "min_ms": lat_ms[0], "max_ms": lat_ms[-1],
                         "pairs_per_s_at_median": n / (lat_ms[len(lat_ms) // 2] * 1e-3)} if lat_ms else None),
            "stages_ms": stages,
            "stages_ms_one_call_at_a_time": iso if iso else None,
            "result_x_prefix": result[:8].hex(),
        }
        if not args.no_cpu_baseline and not sharded:
            out["cpu_baseline"] = cpu_baseline(args, pts, sc_host, n, result)
        print(json.dumps(out), flush=True)
    if prep is not None:
        prep.release()
    pts.free()
    eng.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, pts, sc_host, n, gpu_result):
    """The oracle (C port of the reference algorithm: build_multiexp.js schedule -- c by clz(n), unsigned digits, Jacobian addMixed,
    recursive reduceTable) timed on the host cores of this box, on a bounded sample of the same workload, and used as a parity
    check.  Parallelised exactly as the reference's external caller does it (one `_chunk` task per window handed to a worker pool,
    Horner-combined at the end), on up to 16 threads; the single-thread rate of the same code is reported as well."""
    import concurrent.futures as cf
    from oracle import oracle as orc
    orc.build()
    O = orc.Oracle(args.curve, args.group)
    asz = 2 * O.fsz
    m = min(n, 1 << 20)
    if args.group == 2:
        m = min(m, 1 << 18)
    bases = pts.download(m * asz)
    sc = sc_host[:m].tobytes()
    # the reference's window size for this n (build_multiexp.js:275-280)
    tsizes = [17] * 10 + [16, 16, 15, 14, 13, 13, 12, 11, 10, 9, 8, 7, 7, 6, 5, 4, 3, 2, 1, 1, 1, 1]
    c = tsizes[32 - m.bit_length()]
    nchunks = (256 - 1) // c + 1
    threads = max(1, min(16, os.cpu_count() or 1, nchunks))
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:       # ctypes releases the GIL inside the oracle
        parts = list(ex.map(lambda w: O.multiexp_chunk(bases, sc, 32, m, w * c, c), range(nchunks)))
    acc = O.zero()
    for w in range(nchunks - 1, -1, -1):
        for _ in range(c):
            acc = O.double(acc)
        acc = O.add(acc, parts[w])
    dt = time.perf_counter() - t0
    ref = O.normalize(acc)
    ok = (ref == gpu_result) if m == n else None
    # single thread on a smaller slice of the same inputs (about 2 s)
    m1 = min(m, 1 << 17)
    t1 = time.perf_counter()
    O.multiexp_affine(bases[:m1 * asz], sc[:m1 * 32], 32, m1)
    dt1 = time.perf_counter() - t1
    return {"value": m / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "note": "kind 'port' = oracle/oracle.c, the C restatement of the reference schedule (64-bit limbs, native code): several times FASTER "
                    "than the reference's own WASM, whose rate (BASELINE.md section 2: g1m_multiexpAffine under Node v12, ONE thread of an 8-vCPU "
                    "Intel Xeon @ 2.10 GHz build container, n = 2^16) was 1.06e4 pairs/s; the .wasm cannot travel to this box, so the port is the "
                    "(tougher) baseline timed here",
            "reference_wasm_in_container": {"value": 1.06e4, "unit": "pairs/s", "cores": 1, "n": 65536, "hardware": "Intel Xeon @ 2.10 GHz (build container), Node v12.22.9", "source": "BASELINE.md section 2"},
            "sample": f"first {m} pairs of the same workload: {nchunks} window tasks (c = {c}, the reference's schedule) of the oracle's "
                      f"multiexpAffine_chunk on {threads} threads + Horner combine, {dt:.1f} s",
            "single_thread_value": m1 / dt1, "single_thread_sample": f"first {m1} pairs, one call of the oracle's multiexpAffine, {dt1:.1f} s",
            "matches_gpu_result": ok}


if __name__ == "__main__":
    main()
