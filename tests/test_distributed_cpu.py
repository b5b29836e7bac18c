"""The N > 1 path on CPU: world_size-2, -3 and -4 gloo process groups run the gather + combine protocol of
msm_amd.dist.  There is no GPU here, so each rank's partial -- its window group of its point slice (gp point
slices x gw window groups; gw = world is the windows-only form) -- is produced by the oracle's per-window entry
point (the same decomposition the engine's msm_amd_multiexp_shard computes on a GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, curve, group, n, c, q, gw=None):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import golden_util as gu
    from msm_amd import dist as mdist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, f"dist:{curve}:{group}", n)
        all_bases, all_scalars = bases, scalars
        # rank r = p * gw + w: the windows {k : k % gw == w} of point slice p (msm_amd.dist.slice_of = msm_dist_slice of the engine)
        gp, gw = mdist.split_of(world, gw or world)
        first, count = mdist.slice_of(n, gp, rank // gw)
        asz = 2 * O.fsz
        bases, scalars = bases[first * asz:(first + count) * asz], scalars[first * 32:(first + count) * 32]
        # this rank's windows, weighted by 2^(c*w): Horner from its top window down
        W = (256 + c - 1) // c
        mine = [w for w in range(W) if w % gw == rank % gw]
        acc = O.zero()
        prev = None
        for w in reversed(mine):
            if prev is not None:
                for _ in range((prev - w) * c):
                    acc = O.double(acc)
            acc = O.add(acc, O.multiexp_chunk(bases, scalars, 32, count, w * c, c))
            prev = w
        if prev is not None:
            for _ in range(prev * c):
                acc = O.double(acc)
        total = mdist.gather_and_combine(curve, group, acc)
        # the asynchronous form bench.py uses for N > 1: two exchanges in flight, collected in order
        xch = mdist.PartialExchange(curve, group)
        h1, h2 = xch.submit(acc), xch.submit(O.zero())
        assert xch.result(h1) == total
        assert xch.result(h2) == O.normalize(O.zero())
        if rank == 0:
            exp = O.normalize(O.multiexp_affine(all_bases, all_scalars, 32, n))
            q.put((total == exp, total.hex()[:16]))
        else:
            q.put((True, total.hex()[:16]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,curve,group,c", [(2, "bls12381", 1, 16), (3, "bn128", 1, 13), (2, "bls12381", 2, 16)])
def test_window_sharded_gather_combine_gloo(world, curve, group, c):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, group, 48, c, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for ok, _ in res)
    assert len({h for _, h in res}) == 1      # every rank holds the same combined result


@pytest.mark.parametrize("world,gw,curve,group,c,n", [(4, 2, "bls12381", 1, 16, 50), (2, 1, "bn128", 1, 13, 49), (4, 1, "bn128", 2, 16, 30)])
def test_hybrid_point_slices_and_window_groups_gloo(world, gw, curve, group, c, n):
    """gp point slices x gw window groups (the split msm_amd_multiexp_dist plans per call): 2 x 2 over four ranks, pure point slices
    over two and four (ragged: 49 = 25 + 24, 30 = 8 + 8 + 8 + 6) -- one all_gather of `world` partials, the same sum on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, group, n, c, q, gw)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for ok, _ in res)
    assert len({h for _, h in res}) == 1
