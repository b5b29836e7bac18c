"""Host-side planning of the engine, through the C ABI and without a GPU: the window layout of prepared tables
(msm_amd_plan_prepared = what msm_amd_prepare_bases builds) and the split of a multi-GPU call (msm_amd_dist_plan with no context).
The digit recoding that the layout promises -- `ns` signed windows of c bits, then unsigned ones of c - 1 bits, every magnitude at
most 2^(c-1), no carry out of the top window -- is replayed in Python on random scalars (the device code that does the same,
DigitSource::next in csrc/sort_kernels.hpp, is held to the oracle by the GPU tests)."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zprize-wasm-msm_amd", "python"))

msm_amd = pytest.importorskip("msm_amd")
AFFINE = {("bls12381", 1): 96, ("bls12381", 2): 192, ("bn128", 1): 64, ("bn128", 2): 128}


def _start_bit(c, ns, w):
    return w * c if w <= ns else ns * c + (w - ns) * (c - 1)      # window_start_bit, csrc/engine_common.hpp


def _digits(s, c, W, ns):
    """signed digits of s under the layout (c, W, ns): [(window, digit)], digit in [-2^(c-1), 2^(c-1)]; asserts what the layout promises"""
    half, carry, out = 1 << (c - 1), 0, []
    for w in range(W):
        wb = c if w < ns else c - 1
        raw = ((s >> _start_bit(c, ns, w)) & ((1 << wb) - 1)) + carry
        if w < ns and raw > half:
            out.append((w, raw - (1 << c)))
            carry = 1
        else:
            assert raw <= half, (w, raw)       # an unsigned window holds digit + carry <= 2^(c-1)
            out.append((w, raw))
            carry = 0
    assert carry == 0                          # the top window never spills
    assert s >> _start_bit(c, ns, W) == 0      # and nothing of the scalar lies beyond it
    return out


def test_headline_layout_and_table_size():
    p = msm_amd.plan_prepared("bls12381", 1, 1 << 20)
    assert (p["scalar_bits"], p["c"], p["windows"], p["signed_windows"], p["buckets"]) == (255, 20, 13, 8, 1 << 19)
    assert p["table_bytes"] == 13 * (1 << 20) * 96
    assert msm_amd.plan_prepared("bn128", 1, 1 << 20)["scalar_bits"] == 254
    # a width that 20-bit windows overshoot (7 x 19 = 133 > 128: a thin top window) takes the nearest size that tiles it exactly
    p = msm_amd.plan_prepared("bls12381", 1, 1 << 20, 128)
    assert (p["c"], p["windows"], p["signed_windows"]) == (19, 7, 2) and 2 * 19 + 5 * 18 == 128
    assert msm_amd.plan_prepared("bls12381", 1, 1 << 20, 128, 20)["c"] == 20          # a forced size is a forced size


@pytest.mark.parametrize("curve,group", list(AFFINE))
def test_prepared_layouts_cover_the_scalars_exactly(curve, group):
    rng = random.Random(f"{curve}{group}")
    sizes = [1, 2, 3, 100, 4095, 4096, 4097, 65535, 65536, 200000, (1 << 18) + 1, 1 << 19, 1 << 20, 3 << 20, 1 << 24, 1 << 26]
    for n in sizes:
        for bits in (0, 1, 8, 64, 127, 128, 253, 254, 255, 256, 300, 512):
            for forced in (0, rng.choice([3, 7, 9, 13, 16, 17, 20, 21])):
                try:
                    p = msm_amd.plan_prepared(curve, group, n, bits, forced)
                except msm_amd.MsmAmdError as e:
                    # refused only when no window size keeps windows x (padded) points below 2^31
                    assert e.code == msm_amd.EINVAL
                    stride = (n + 8191) & ~8191
                    assert ((bits or 255) + 21) // 21 * stride >= 1 << 31, (n, bits, forced)
                    continue
                c, W, ns, sb = p["c"], p["windows"], p["signed_windows"], p["scalar_bits"]
                assert sb == (bits or (255 if curve == "bls12381" else 254))
                assert 3 <= c <= 21 and p["buckets"] == 1 << (c - 1)
                assert W == (sb + c) // c and 0 <= ns <= max(W - 1, 0)          # the fewest windows; the top one unsigned
                assert _start_bit(c, ns, W) >= sb                              # they cover the scalar ...
                assert ns == 0 or _start_bit(c, ns - 1, W) < sb                # ... with the fewest signed ones
                stride = (n + 8191) & ~8191
                assert W * stride < 1 << 31 and p["table_bytes"] == W * stride * AFFINE[(curve, group)]
                if forced and W * stride < 1 << 31:
                    assert c >= forced                                         # a forced size is kept unless the table would not fit
                for s in [0, 1, (1 << sb) - 1, 1 << (sb - 1)] + [rng.getrandbits(sb) for _ in range(6)] + \
                         [sum(1 << (c - 1) << _start_bit(c, ns, w) for w in range(W) if _start_bit(c, ns, w) + c - 1 < sb)]:
                    d = _digits(s, c, W, ns)
                    assert sum(v << _start_bit(c, ns, w) for w, v in d) == s


def test_prepared_plan_refuses_bad_arguments():
    for args in [("bls12381", 3, 10), ("bls12381", 1, 0), ("bls12381", 1, 10, 513)]:
        with pytest.raises(msm_amd.MsmAmdError):
            msm_amd.plan_prepared(*args)
    with pytest.raises(msm_amd.MsmAmdError):
        msm_amd.plan_prepared("bls12381", 1, 1 << 28)        # 13 windows x 2^28 points: beyond 31-bit table indices at any window size


@pytest.mark.parametrize("world", [1, 2, 3, 4, 6, 8, 16])
def test_split_is_a_factorisation_and_the_slices_tile_the_points(world):
    for n in (0, 1, 5, 1000, 4097, 65536, 200001, 1 << 20, (1 << 22) + 3, 1 << 26):
        plans = [msm_amd.plan_split("bls12381", 1, 32, n, world, r) for r in range(world)]
        gp, gw = plans[0]["gp"], plans[0]["gw"]
        assert gp * gw == world and all((p["gp"], p["gw"]) == (gp, gw) for p in plans)
        # rank r = p * gw + w: the gw ranks of a slice agree on it, consecutive slices tile [0, n)
        pos = 0
        for p in range(gp):
            sl = {(plans[p * gw + w]["first"], plans[p * gw + w]["count"]) for w in range(gw)}
            assert len(sl) == 1
            first, count = sl.pop()
            assert first == pos
            pos += count
        assert pos == n
        counts = [plans[p * gw]["count"] for p in range(gp)]
        per = (n + gp - 1) // gp               # msm_dist_slice: slices of ceil(n / gp) points, the last one takes what is left
        assert all(k == per for k in counts[:-1]) and 0 <= counts[-1] <= per   # slice 0 is (one of) the longest
        if gp > 1:
            assert per >= 1024                 # slices of fewer than 1024 points are all fixed cost: such a call is not cut along the points


def test_planned_splits_of_the_measured_table():
    """the plan follows profiles/r04_shard_per_rank_table.txt (busiest rank's share, measured on one MI355X with round 4's kernels)"""
    split = lambda logn, world: (lambda p: (p["gp"], p["gw"]))(msm_amd.plan_split("bls12381", 1, 32, 1 << logn, world, 0))
    assert split(26, 8) == (8, 1) and split(26, 2) == (2, 1)       # large n: along the points, every rank holds 1/8 of the bases
    assert split(20, 2) == (2, 1) and split(20, 4) == (1, 4) and split(20, 8) == (4, 2)   # the headline size: both seams, by the measured shares
    assert split(16, 8)[0] * split(16, 8)[1] == 8


def test_split_refuses_bad_arguments():
    for args in [(32, 10, 4, 4), (0, 10, 4, 0), (65, 10, 4, 0)]:
        with pytest.raises(msm_amd.MsmAmdError):
            msm_amd.plan_split("bls12381", 1, *args)
    with pytest.raises(msm_amd.MsmAmdError):
        msm_amd.plan_split("bls12381", 1, 32, 10, 0, 0)
