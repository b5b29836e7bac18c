"""Multi-GPU MSM: point slices x Pippenger windows sharded over the ranks of one node (one process per GPU).

The reference already exposes one window as an independently callable unit (`fnName_chunk`,
wasmcurves/src/build_multiexp.js:96-104, exported :469); its external caller fans windows out to
workers and adds the results -- and cuts inputs beyond 2^22 points into slices, the MSM being linear
in them.  Here the ranks form gp point slices x gw window groups: rank r = p * gw + w computes the
windows {k : k % gw == w} of slice p -- each already weighted by 2^(c*k) -- as one Jacobian partial
(msm_amd_multiexp_shard), the partials (144 B for G1, 288 B for G2) are exchanged with ONE
all_gather (RCCL over xGMI when the backend is "nccl"), and every rank adds them on the host
(msm_amd_combine).  Elliptic-curve addition is not an RCCL reduction operator, hence gather + add.
"""
import torch
import torch.distributed as dist

from . import combine


def _start_gather(partial, device, process_group, async_op):
    world = dist.get_world_size(process_group)
    mine = torch.frombuffer(bytearray(partial), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    flat = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    work = dist.all_gather_into_tensor(flat, mine, group=process_group, async_op=async_op)
    return work, flat, mine


def gather_and_combine(curve, group, partial, device=None, process_group=None):
    """all_gather one Jacobian partial per rank (one collective into one flat tensor, one copy back) and return the
    normalised sum (same bytes on every rank)."""
    _, flat, _ = _start_gather(partial, device, process_group, False)
    return combine(curve, group, flat.cpu().numpy().tobytes())


class PartialExchange:
    """The same exchange for a caller that keeps several sharded calls in flight: `submit` starts the all_gather of one
    call's partial and returns at once; `result` waits for it, copies the world x 144 B back and adds them on the host.
    The collective of call k then runs while the host collects the partial of call k+1 (bench.py, N > 1)."""

    def __init__(self, curve, group, device=None, process_group=None):
        self.curve, self.group, self.device, self.pg = curve, group, device, process_group

    def submit(self, partial):
        return _start_gather(partial, self.device, self.pg, True)

    def result(self, handle):
        work, flat, _mine = handle
        work.wait()
        return combine(self.curve, self.group, flat.cpu().numpy().tobytes())


def split_of(world, gw):
    """(gp, gw) of a world of ranks cut into gw window groups: rank r = p * gw + w"""
    assert world % gw == 0
    return world // gw, gw


def slice_of(n, gp, p):
    """points [first, first + count) of slice p of gp (msm_dist_slice in csrc/engine_ctx.hpp)"""
    per = (n + gp - 1) // gp
    lo = min(p * per, n)
    return lo, min(lo + per, n) - lo


def multiexp_sharded(curve_obj, group, bases, scalars, scalar_size, n, device=None, process_group=None, jacobian_bases=False, gw=None):
    """R = sum_i scalar_i * base_i split over the process group: gp point slices x gw window groups (gw = None: windows only, the
    form of rounds 1-2; the C-ABI path msm_amd_multiexp_dist picks gw itself).  `curve_obj` is an msm_amd.Curve bound to this rank's
    GPU; bases/scalars are this rank's copies of ALL n points (host bytes, or DeviceBuffers whose slices are taken as views)."""
    rank = dist.get_rank(process_group)
    world = dist.get_world_size(process_group)
    gp, gw = split_of(world, gw or world)
    first, count = slice_of(n, gp, rank // gw)
    stride = (curve_obj.jacobian_bytes(group) if jacobian_bases else curve_obj.affine_bytes(group))
    if hasattr(bases, "view") and hasattr(bases, "ptr"):
        b, s = bases.view(first * stride, count * stride), scalars.view(first * scalar_size, count * scalar_size)
    else:
        flat = lambda x: x.reshape(-1) if hasattr(x, "reshape") else x      # torch / numpy inputs of any shape: a flat byte range
        b, s = flat(bases)[first * stride:(first + count) * stride], flat(scalars)[first * scalar_size:(first + count) * scalar_size]
    part = curve_obj.multiexp_shard(group, b, s, scalar_size, count, rank % gw, gw, jacobian_bases)
    return gather_and_combine(curve_obj.name, group, part, device, process_group)
