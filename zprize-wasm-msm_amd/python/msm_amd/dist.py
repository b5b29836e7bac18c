"""Multi-GPU MSM: Pippenger windows sharded over the ranks of one node (one process per GPU).

The reference already exposes one window as an independently callable unit (`fnName_chunk`,
wasmcurves/src/build_multiexp.js:96-104, exported :469); its external caller fans windows out to
workers and adds the results.  Here rank r computes the windows {w : w % world == r} of the SAME
(bases, scalars) -- each already weighted by 2^(c*w) -- as one Jacobian partial
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


def multiexp_sharded(curve_obj, group, bases, scalars, scalar_size, n, device=None, process_group=None, jacobian_bases=False):
    """R = sum_i scalar_i * base_i with the windows sharded over the process group.
    `curve_obj` is an msm_amd.Curve bound to this rank's GPU; bases/scalars are this rank's copies."""
    rank = dist.get_rank(process_group)
    world = dist.get_world_size(process_group)
    part = curve_obj.multiexp_shard(group, bases, scalars, scalar_size, n, rank, world, jacobian_bases)
    return gather_and_combine(curve_obj.name, group, part, device, process_group)
