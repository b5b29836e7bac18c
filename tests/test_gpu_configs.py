"""GPU tests added in round 2 (run with -m gpu on the MI355X box):
  * the BASELINE configurations at their stated sizes (config 3: BLS12-381 G1 n = 2^26; config 5: BN254 G1 n = 2^24; config 4:
    BLS12-381 G2 n = 2^20) through size-independent properties + an oracle spot slice -- they cross thresholds smaller cases do
    not (32-bit positions, multi-GiB scratch, wide windows, deep buckets);
  * bases outside the order-r subgroup (the reference's multiexp is exact for any point of the curve);
  * the guard on 32-bit (pair, window) positions;
  * `bench.py --gpus 2` end to end (ranks started by bench.py itself; gloo on a one-GPU box, RCCL when two GPUs are present);
  * two ranks on two GPUs exchanging their window partials over RCCL (skipped on a one-GPU box)."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    import msm_amd
    e = msm_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc(oracle_lib):
    return oracle_lib


def _call(eng, C, group, bases_ptr, scalars_ptr, n, flags=0):
    """msm_amd_multiexp on raw device addresses (slices of one resident buffer: plain pointer arithmetic)"""
    import msm_amd
    out = ctypes.create_string_buffer(C.jacobian_bytes(group))
    eng._chk(eng.L.msm_amd_multiexp(eng.ctx, C.cid, group, msm_amd.FLAG_DEVICE | flags, bases_ptr, scalars_ptr, 32, n, out))
    return out.raw


@pytest.mark.parametrize("curve,group,logn", [("bls12381", 1, 26), ("bn128", 1, 24), ("bls12381", 2, 20)])
def test_baseline_configs_at_full_size(eng, orc, curve, group, logn):
    """BASELINE.json configs 3, 5 and 4 at their stated sizes on one GPU: the eight window shards of the multi-GPU decomposition add
    up to the full result; two point slices add up to it (MSM is linear in the points); the scalar halves k = lo + 2^128 hi do;
    a 2^11 slice of the very same resident inputs equals the oracle."""
    import torch
    import msm_amd
    n = 1 << logn
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    asz = C.affine_bytes(group)
    pts = C.gen_points(group, 0xC0F16 + logn, n)
    # scalars on the GPU (2 GiB at n = 2^26: never staged through numpy), 255 significant bits
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + logn)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    sc[:, 31] &= 0x7F
    torch.cuda.synchronize()
    sp = sc.data_ptr()
    try:
        full = _call(eng, C, group, pts.ptr, sp, n)
        geom = eng.geometry()
        assert geom["entries"] > n * (geom["windows"] - 2)     # (nearly) every window of every pair was scattered
        # (1) the multi-GPU decompositions on one GPU: eight window shards of all the points; the hybrid form the engine plans for
        #     eight ranks (gp point slices x gw window groups, every rank on a view of its slice only)
        parts = b"".join(C.multiexp_shard(group, pts, sc, 32, n, r, 8) for r in range(8))
        assert msm_amd.combine(curve, group, parts) == full
        from msm_amd import dist as mdist
        plan = eng.dist_plan(curve, group, 32, n, 8, 0)
        hybrid = []
        for r in range(8):
            first, count = mdist.slice_of(n, plan["gp"], r // plan["gw"])
            hybrid.append(C.multiexp_shard(group, pts.view(first * asz, count * asz), sc.reshape(-1)[first * 32:(first + count) * 32], 32, count,
                                           r % plan["gw"], plan["gw"]))
        assert msm_amd.combine(curve, group, b"".join(hybrid)) == full
        if plan["gp"] != 2:          # and a split that cuts both ways
            hybrid = []
            for r in range(8):
                first, count = mdist.slice_of(n, 2, r // 4)
                hybrid.append(C.multiexp_shard(group, pts.view(first * asz, count * asz), sc.reshape(-1)[first * 32:(first + count) * 32], 32, count, r % 4, 4))
            assert msm_amd.combine(curve, group, b"".join(hybrid)) == full
        # (2) point slices: uneven cut, so the two halves see different n (and possibly different window sizes)
        cut = (n // 3) | 1
        lo = _call(eng, C, group, pts.ptr, sp, cut)
        hi = _call(eng, C, group, pts.ptr + cut * asz, sp + cut * 32, n - cut)
        assert msm_amd.combine(curve, group, lo + hi) == full
        # (3) oracle spot slice of the same resident inputs
        m = 2048
        h_pts = pts.download(m * asz)
        h_sc = sc[:m].cpu().numpy().tobytes()
        exp = O.normalize(O.multiexp_affine(h_pts, h_sc, 32, m))
        assert _call(eng, C, group, pts.ptr, sp, m) == exp
        # (4) the _chunk entry point at full size: bits [0, 128) and [128, 256) through 25-bit chunks would take 11 calls; take the
        #     two halves of the scalar instead: sum_i (k_i mod 2^128) P_i + 2^128 sum_i (k_i >> 128) P_i == full
        lo128 = sc.clone()
        lo128[:, 16:] = 0
        hi128 = torch.zeros_like(sc)
        hi128[:, :16] = sc[:, 16:]
        torch.cuda.synchronize()
        a = _call(eng, C, group, pts.ptr, lo128.data_ptr(), n)
        b = _call(eng, C, group, pts.ptr, hi128.data_ptr(), n)
        acc = b
        for _ in range(128):
            acc = O.double(acc)
        assert O.normalize(O.add(acc, a)) == full
    finally:
        pts.free()
        del sc
        torch.cuda.empty_cache()


def test_bases_outside_the_subgroup_match_the_reference(eng, orc):
    """The reference's g1m_multiexpAffine is integer arithmetic on curve points: exact for points outside the order-r subgroup too.
    The engine's GLV schedule (k mod r, endomorphism) is not -- so it only runs for callers that pass MSM_AMD_BASES_IN_SUBGROUP.
    n = 200000 sits in the range (180000 <= n <= 2^18) where the GLV schedule is the default for flagged calls (unflagged ones run 19
    windows of 14 / 13 bits there); a smaller n = 4500 (8-bit windows, no GLV schedule whatever the flag says) is checked as well."""
    import msm_amd
    curve, group, n = "bls12381", 1, 200000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    with open(os.path.join(gu.GOLDEN, "golden_ingroup.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["curve"] == curve and c["group"] == group]
    outside = []
    for c in cases:
        aff = bytes.fromhex(c["affine_in"])
        for i, (ing, inc) in enumerate(zip(c["inGroupAffine"], c["inCurveAffine"])):
            if inc and not ing:
                outside.append(aff[96 * i:96 * (i + 1)])
    assert outside, "the fixture holds on-curve points outside the subgroup"
    bases, scalars = gu.random_case(curve, group, "gpu:cofactor", n)
    bases = bytearray(bases)
    for i in range(0, n, 3):                                   # every third base is a cofactor point
        bases[96 * i:96 * (i + 1)] = outside[(i // 3) % len(outside)]
    bases = bytes(bases)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp               # default: no assumption about the bases
    assert (eng.geometry()["c"], eng.geometry()["windows"]) == (14, 20)     # windows of even width over the 255 bits of a reduced scalar (8 of 14 bits, 11 of 13) + the overflow window
    # flagged calls take the GLV schedule; with bases that really are in the subgroup it gives the reference's bytes
    good, gs = gu.random_case(curve, group, "gpu:cofactor:good", n)
    expg = O.normalize(O.multiexp_affine(good, gs, 32, n))
    assert C.g1m_multiexpAffine(good, gs, 32, n, in_subgroup=True) == expg
    assert eng.geometry()["windows"] == 8
    assert C.g1m_multiexpAffine(good, gs, 32, n) == expg
    m = 4500
    exps = O.normalize(O.multiexp_affine(bases[:96 * m], scalars[:32 * m], 32, m))
    for flag in (False, True):                                  # the flag is the caller's word about the bases; here it changes nothing
        assert C.g1m_multiexpAffine(bases[:96 * m], scalars[:32 * m], 32, m, in_subgroup=flag) == exps
        assert eng.geometry()["windows"] == 33


@pytest.mark.parametrize("curve,group", [("bls12381", 1), ("bn128", 2)])
def test_wide_windows_and_both_forms_of_the_fine_sort(eng, orc, curve, group):
    """Windows wider than 16 bits (8 coarse + up to 12 fine bits of bucket index, 8-byte partition entries are not needed at this n)
    and the two forms of the fine sort -- one workgroup per span, or long spans cut into chunks sorted in parallel (option
    "sort_big": 1 = always the chunk-parallel kernels, 2 = never) -- all give the oracle's bytes.  n = 12000 with 9-bit windows
    makes every span long (12000 / 1 coarse bin), with 17..21-bit windows the spans are short and mostly empty."""
    n = 12000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, f"gpu:wide:{curve}:{group}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    f = getattr(C, f"g{group}m_multiexpAffine")
    try:
        for c in (9, 17, 19, 21):
            eng.set_window_bits(c)
            for big in (1, 2, 0):
                eng.set_option("sort_big", big)
                assert f(bases, scalars, 32, n) == exp, (c, big)
                assert eng.geometry()["c"] == c
        # a window shard and a _chunk call on the wide path
        eng.set_window_bits(18)
        import msm_amd
        parts = b"".join(C.multiexp_shard(group, bases, scalars, 32, n, r, 3) for r in range(3))
        assert msm_amd.combine(curve, group, parts) == exp
    finally:
        eng.set_window_bits(0)
        eng.set_option("sort_big", 0)


def test_all_equal_scalars_take_the_long_span_paths(eng, orc):
    """Adversarial scalars (all equal: one bucket per window receives every point) at a size where that bucket's span is far beyond
    what one workgroup reads at once: the chunk-parallel sort, the single-workgroup fallback and the segmented accumulation of
    over-long buckets all have to agree with n * (k * P) computed by the oracle."""
    curve, group, n = "bls12381", 1, 50000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, _ = gu.random_case(curve, group, "gpu:alleq", 16)
    k = (0x1234567890abcdef1122334455667788 << 96 | 0x8000_0000_7fff_8001).to_bytes(32, "little")
    scalars = k * n
    big_bases = (bases * (n // 16 + 1))[:96 * n]
    # sum_i k * P_(i mod 16) = k * (n/16 * sum of the 16 points) + remainder: let the oracle do it on the 16 distinct points
    cnt = [n // 16 + (1 if i < n % 16 else 0) for i in range(16)]
    sc16 = b"".join(((int.from_bytes(k, "little") * c) % gu.R_ORDER[curve]).to_bytes(32, "little") for c in cnt)
    exp = O.normalize(O.multiexp_affine(bases, sc16, 32, 16))
    try:
        for big in (1, 2, 0):
            eng.set_option("sort_big", big)
            assert C.g1m_multiexpAffine(big_bases, scalars, 32, n) == exp, big
    finally:
        eng.set_option("sort_big", 0)


@pytest.mark.parametrize("curve", ["bls12381", "bn128"])
def test_scalar_widths_that_leave_dense_or_carry_buckets(eng, orc, curve):
    """Scalars are plain 32-byte integers to the reference (build_multiexp.js:36-94 slices bits, nothing reduces mod r).  Widths that
    are not the group order's make some buckets far denser than the rest: 256 bits -- half of the scalars carry out of the top 16-bit
    window into a 17th whose single bucket then holds n/2 entries; 250 or 243 bits -- a narrow top window, few buckets, each 2^k
    times the mean.  Those buckets are cut into segments (tail kernels: one lane per bucket for a few segments, a workgroup for many)
    and, with a narrow forced window, the launch is sliced as well: every combination has to match the oracle."""
    import random
    group, n = 1, 20000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, _ = gu.random_case(curve, group, f"gpu:widths:{curve}", n)
    rnd = random.Random(f"widths:{curve}")
    try:
        for bits in (256, 250, 243):
            scalars = b"".join((rnd.getrandbits(bits) | (rnd.getrandbits(1) << (bits - 1))).to_bytes(32, "little") for _ in range(n))
            exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
            for c in (0, 11):
                eng.set_window_bits(c)
                assert C.g1m_multiexpAffine(bases, scalars, 32, n) == gu.canon(curve, group, exp, O.zero()), (curve, bits, c)
    finally:
        eng.set_window_bits(0)


def test_blocked_fft_building_blocks_match_the_reference(eng, orc):
    """<p>_fftJoin / _fftJoinExt / _fftJoinExtInv / _prepareLagrangeEvaluation / _fftMix / _fftFinal for p = frm (byte for byte), g1m, g2m
    (as normalised points) and frm_batchTo/FromMontgomery == the reference WASM (tests/golden/golden_fft2.json, oracle/gen_golden_fft2.js).
    Size-independent check on top: a 2^12-point Fr transform assembled from its fftMix passes equals frm_fft of the bit-reversed input."""
    with open(os.path.join(gu.GOLDEN, "golden_fft2.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        C = eng.curve(c["curve"])
        g = c["group"]
        if g == -1:
            plain, mont = bytes.fromhex(c["plain"]), bytes.fromhex(c["mont"])
            assert C.frm_batchToMontgomery(plain) == mont
            assert C.frm_batchFromMontgomery(mont) == bytes.fromhex(c["back"]) == plain
            continue
        in1, in2 = bytes.fromhex(c["in1"]), bytes.fromhex(c["in2"])
        first, inc, factor = (bytes.fromhex(c[k]) for k in ("first", "inc", "factor"))
        if g:
            O = orc.Oracle(c["curve"], g)
            zero, jsz = O.zero(), O.jsz

            def canon(hx):
                b = bytes.fromhex(hx)
                return b"".join(gu.canon(c["curve"], g, b[i:i + jsz], zero) for i in range(0, len(b), jsz))
        else:
            def canon(hx):
                return bytes.fromhex(hx)
        for name, exp in c["ops"].items():
            op, _, arg = name.partition(":")
            tag = (c["curve"], g, c["n"], name)
            if op == "fftJoin":
                got = C.fft_join(g, in1, in2, first, inc)
            elif op == "fftJoinExt":
                got = C.fft_join_ext(g, in1, in2, first, inc, int(arg))
            elif op == "fftJoinExtInv":
                got = C.fft_join_ext(g, in1, in2, first, inc, int(arg), inverse=True)
            elif op == "prepareLagrangeEvaluation":
                got = C.prepare_lagrange_evaluation(g, in1, in2, first, inc, int(arg))
            elif op == "fftMix":
                got = (C.fft_mix(g, in1, int(arg)),)
            elif arg == "odd3":
                got = (C.fft_final(g, in1[:3 * 32], factor),)
            else:
                got = (C.fft_final(g, in1, factor),)
            assert got[0] == canon(exp["out1"]), tag
            if "out2" in exp:
                assert got[1] == canon(exp["out2"]), tag
    # fftMix passes 1..bits over the bit-reversed input ARE the radix-2 transform
    for curve in ("bls12381", "bn128"):
        C = eng.curve(curve)
        bits, r = 12, gu.R_ORDER[curve]
        raw = gu.stream(f"gpu:fftmix:{curve}", 40 << bits)
        x = [(int.from_bytes(raw[40 * i:40 * i + 40], "little") % r).to_bytes(32, "little") for i in range(1 << bits)]
        buf = b"".join(x[int(format(i, f"0{bits}b")[::-1], 2)] for i in range(1 << bits))
        for e in range(1, bits + 1):
            buf = C.fft_mix(0, buf, e)
        assert buf == C.frm_fft(b"".join(x))


@pytest.mark.parametrize("curve", ["bls12381", "bn128"])
def test_lane_pair_reduction_matches_oracle(eng, orc, curve):
    """Option reduce_pair = 1: the G1 bucket reduction with every point split over a lane pair (XYZZH: even lane (x, zz), odd lane
    (y, zzz); eight product rounds per addition).  Random inputs, repeated points (the doubling branch of the pair addition: equal
    buckets meet in the running sums), P and -P, infinity among the bases, several window sizes and L."""
    n = 6000
    C = eng.curve(curve)
    O = orc.Oracle(curve, 1)
    asz = 2 * O.fsz
    bases, scalars = gu.random_case(curve, 1, f"gpu:pairred:{curve}", n)
    bases = bytearray(bases)
    bases[asz * 7:asz * 8] = bytes(asz)                                   # affine infinity
    for i in range(100, 200):                                             # one point a hundred times ...
        bases[asz * i:asz * (i + 1)] = bases[asz * 100:asz * 101]
    bases = bytes(bases)
    sc = bytearray(scalars)
    for i in range(100, 200):                                             # ... with scalars 1, 2, 3, ...: neighbouring buckets hold equal sums
        sc[32 * i:32 * (i + 1)] = (i - 99).to_bytes(32, "little")
    scalars = bytes(sc)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    try:
        eng.set_option("reduce_pair", 1)
        for c, L in ((0, 0), (16, 1), (16, 32), (9, 4), (4, 0)):
            eng.set_window_bits(c)
            eng.set_option("reduce_l", L)
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp, (c, L)
    finally:
        eng.set_option("reduce_pair", 0)
        eng.set_option("reduce_l", 0)
        eng.set_window_bits(0)


def test_reduce_l_must_be_a_power_of_two(eng):
    """The LDS trees of the bucket reduction halve the lane count: a lane count that is not a power of two would drop a partial
    sum (reduce_l = 200 at c = 16 gave 164 lanes).  The option refuses such values instead."""
    import msm_amd
    with pytest.raises(msm_amd.MsmAmdError, match="power of two"):
        eng.set_option("reduce_l", 200)
    eng.set_option("reduce_l", 64)
    eng.set_option("reduce_l", 0)


def test_entries_of_one_pass_must_fit_32_bits(eng):
    """windows x points >= 2^32 would wrap the 32-bit positions of the sort: refused, not computed wrongly.  One-bit windows over
    256-bit scalars are 257 windows; 2^24 points x 257 > 2^32.  (Buffers are uninitialised: the call is refused before any kernel.)"""
    import msm_amd
    n = 1 << 24
    C = eng.curve("bn128")
    b, s = eng.alloc(n * 64), eng.alloc(n * 32)
    try:
        eng.set_window_bits(1)
        with pytest.raises(msm_amd.MsmAmdError, match="2\\^32"):
            C.multiexp_async(1, b, s, 32, n)
        eng.set_window_bits(0)
        # the context stays usable
        bases, scalars = gu.random_case("bn128", 1, "gpu:guard", 64)
        assert len(C.g1m_multiexpAffine(bases, scalars, 32, 64)) == 96
    finally:
        eng.set_window_bits(0)
        b.free()
        s.free()


def test_bench_gpus_2_reports_two_ranks():
    """`python bench.py --gpus 2` as the driver types it (no launcher around it): bench.py starts the two ranks itself.  With two GPUs
    the exchange is RCCL; on a one-GPU box the same flow is rehearsed with gloo, both ranks on the one GPU."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if torch.cuda.device_count() < 2:
        env["MSM_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--logn", "16"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0
    assert "over 2 GPUs" in line["config"]["parallelism"]
    assert line["config"]["split"]["gp"] * line["config"]["split"]["gw"] == 2


def test_dist_entry_points_with_a_group_of_one(orc):
    """The exchange behind the C ABI (msm_amd_dist_init / msm_amd_multiexp_dist[_async] / msm_amd_dist_wait: RCCL communicator,
    ncclAllGather of the partial, host combine) with the one rank a one-GPU box can hold."""
    import msm_amd
    curve, group, n = "bls12381", 1, 3000
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, "gpu:dist1", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        with pytest.raises(msm_amd.MsmAmdError, match="dist_init"):
            C.multiexp_dist(group, bases, scalars, 32, n)
        eng.dist_init(0, 1, msm_amd.dist_unique_id())
        assert C.multiexp_dist(group, bases, scalars, 32, n) == exp
        t1 = C.multiexp_dist_async(group, bases, scalars, 32, n)
        t2 = C.multiexp_dist_async(group, bases, scalars, 32, n)
        assert t1.wait() == exp and t2.wait() == exp
        with pytest.raises(msm_amd.MsmAmdError, match="already"):
            eng.dist_init(0, 1, msm_amd.dist_unique_id())
        eng.dist_destroy()
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp      # the context is an ordinary one again
    finally:
        eng.close()


def _cabi_dist_rank(rank, world, q_id, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import msm_amd
    from oracle import oracle as orcm
    eng = msm_amd.Engine(rank)
    try:
        if rank == 0:
            uid = msm_amd.dist_unique_id()
            for _ in range(world - 1):
                q_id.put(uid)
        else:
            uid = q_id.get(timeout=300)
        eng.dist_init(rank, world, uid)
        import threading
        eng.set_option("slots", 4)
        ok = True
        # (1) every rank holds everything (msm_amd_multiexp_dist): the split is planned per call -- points at this size
        curve, group, n = "bn128", 1, 6000
        C = eng.curve(curve)
        bases, scalars = gu.random_case(curve, group, "gpu:cabi-dist", n)
        d_b = eng.alloc(len(bases)).upload(bases)
        d_s = eng.alloc(len(scalars)).upload(scalars)
        r1 = C.multiexp_dist(group, d_b, d_s, 32, n)
        t = [C.multiexp_dist_async(group, d_b, d_s, 32, n) for _ in range(2)]
        r2, r3 = t[0].wait(), t[1].wait()
        ok = ok and r1 == r2 == r3
        # (2) every rank holds only ITS slice (msm_amd_dist_plan + msm_amd_multiexp_dist_local), both forced splits of two ranks
        asz = C.affine_bytes(group)
        for gw in (1, 2):
            eng.set_option("dist_gw", gw)
            plan = eng.dist_plan(curve, group, 32, n)
            assert plan["gw"] == gw and plan["gp"] == world // gw
            lb = bases[plan["first"] * asz:(plan["first"] + plan["count"]) * asz]
            ls = scalars[plan["first"] * 32:(plan["first"] + plan["count"]) * 32]
            ok = ok and C.multiexp_dist_local(group, lb, ls, 32, plan["count"], n) == r1
        eng.set_option("dist_gw", 0)
        # (3) a G1 and a G2 call in flight together, waited for from two threads in the opposite order: the exchanges still pair up
        #     call by call on both ranks (submission order), 144-byte and 288-byte partials never meet
        C2 = eng.curve("bls12381")
        b1, s1 = gu.random_case("bls12381", 1, "gpu:cabi-dist:g1", 3000)
        b2, s2 = gu.random_case("bls12381", 2, "gpu:cabi-dist:g2", 500)
        tk = [C2.multiexp_dist_async(1, b1, s1, 32, 3000), C2.multiexp_dist_async(2, b2, s2, 32, 500), C2.multiexp_dist_async(1, b1, s1, 32, 3000)]
        got = [None] * 3

        def waiter(k):
            got[k] = tk[k].wait()

        th = [threading.Thread(target=waiter, args=(k,)) for k in (2, 1, 0)]
        for x in th:
            x.start()
        for x in th:
            x.join(timeout=300)
        ok = ok and got[0] is not None and got[0] == got[2] and got[1] is not None
        # (4) a rank whose part fails: rank 1 passes a slice length that contradicts the plan -- it gets its own error, rank 0 MSM_AMD_EPEER,
        #     nobody hangs, and the next call is fine
        plan = eng.dist_plan(curve, group, 32, n)
        lb = bases[plan["first"] * asz:(plan["first"] + plan["count"]) * asz]
        ls = scalars[plan["first"] * 32:(plan["first"] + plan["count"]) * 32]
        try:
            C.multiexp_dist_local(group, lb, ls, 32, plan["count"] - (1 if rank == 1 else 0), n)
            ok = False
        except msm_amd.MsmAmdError as e:
            ok = ok and e.code == (msm_amd.EINVAL if rank == 1 else msm_amd.EPEER)
        ok = ok and C.multiexp_dist_local(group, lb, ls, 32, plan["count"], n) == r1
        if rank == 0:
            orcm.build()
            O = orcm.Oracle(curve, group)
            ok = ok and r1 == O.normalize(O.multiexp_affine(bases, scalars, 32, n))
            O1, O2 = orcm.Oracle("bls12381", 1), orcm.Oracle("bls12381", 2)
            ok = ok and got[0] == O1.normalize(O1.multiexp_affine(b1, s1, 32, 3000)) and got[1] == O2.normalize(O2.multiexp_affine(b2, s2, 32, 500))
        q.put((rank, ok, r1.hex()[:16]))
    finally:
        eng.close()


def test_two_ranks_through_the_c_entry_points():
    """The same two-GPU exchange with nothing but the C ABI in the data path: msm_amd_dist_unique_id on rank 0, msm_amd_dist_init on
    both, msm_amd_multiexp_dist[_async] -- what a Node host reaches through the addon's distInit / multiexpDist -- then the hybrid form
    with slice-local inputs (msm_amd_dist_plan / msm_amd_multiexp_dist_local, both splits of two ranks), a G1 and a G2 call in flight
    waited for in the opposite order from two threads, and a call that fails on one rank only (that rank: its own error; the other:
    MSM_AMD_EPEER; the group goes on)."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q, q_id = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_cabi_dist_rank, args=(r, 2, q_id, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert len({h for _, _, h in res}) == 1


def _rccl_rank(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd import dist as mdist
    from oracle import oracle as orcm
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
    eng = msm_amd.Engine(rank)
    try:
        curve, group, n = "bls12381", 1, 5000
        C = eng.curve(curve)
        bases, scalars = gu.random_case(curve, group, "gpu:rccl2", n)
        d_b = eng.alloc(len(bases)).upload(bases)
        d_s = eng.alloc(len(scalars)).upload(scalars)
        # this rank's windows on ITS OWN GPU, two calls in flight, partials exchanged over RCCL/xGMI
        t1 = C.multiexp_shard_async(group, d_b, d_s, 32, n, rank, world)
        t2 = C.multiexp_shard_async(group, d_b, d_s, 32, n, rank, world)
        xch = mdist.PartialExchange(curve, group, device=dev)
        h1, h2 = xch.submit(t1.wait()), xch.submit(t2.wait())
        r1, r2 = xch.result(h1), xch.result(h2)
        r3 = mdist.multiexp_sharded(C, group, d_b, d_s, 32, n, device=dev)
        ok = r1 == r2 == r3
        if rank == 0:
            orcm.build()
            O = orcm.Oracle(curve, group)
            ok = ok and r1 == O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        q.put((rank, ok, r1.hex()[:16]))
    finally:
        eng.close()
        dist.destroy_process_group()


def test_two_ranks_two_gpus_over_rccl():
    """The N > 1 path for real: one process per GPU, each runs its own window shard through msm_amd_multiexp_shard_async on its own
    MI355X, the 144-byte partials cross xGMI in ONE RCCL all_gather, every rank adds them on the host."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=_rccl_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert len({h for _, _, h in res}) == 1
