"""GPU tests added in round 3 (run with -m gpu on the MI355X box), all through the C ABI:
  * the hybrid split of a multi-GPU call -- gp point slices x gw window groups -- summed on one GPU, for every factorisation of
    2, 3, 4, 6 and 8 ranks, ragged slices included; the split the engine itself plans (msm_amd_dist_plan);
  * the exchange of a group: tickets waited for in any order and from several threads still exchange in submission order,
    G1 and G2 calls mixed; a call that fails on this rank completes with its own error and leaves the sequence intact;
  * window digits of exactly 2^(c-1) at c = 20 and 21 (the packed (bucket, rank) word of the coarse partition);
  * the sort's bin histogram growing between calls of one context (small-bin calls first, then a call with more bins)."""
import ctypes
import os
import threading

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import msm_amd
    e = msm_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc(oracle_lib):
    return oracle_lib


def _hybrid_parts(C, group, bases, scalars, n, gp, gw, asz):
    """the world = gp * gw partials of the hybrid decomposition, computed one after the other on this GPU"""
    from msm_amd import dist as mdist
    parts = []
    for r in range(gp * gw):
        first, count = mdist.slice_of(n, gp, r // gw)
        parts.append(C.multiexp_shard(group, bases[first * asz:(first + count) * asz], scalars[first * 32:(first + count) * 32], 32, count, r % gw, gw))
    return b"".join(parts)


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 5000), ("bn128", 1, 130000), ("bls12381", 2, 1500), ("bls12381", 1, 7)])
def test_hybrid_shards_sum_to_full(eng, orc, curve, group, n):
    """Every (gp, gw) with gp * gw in {2, 3, 4, 6, 8}: the partials of the point slices x window groups add up to the oracle's result
    (slices of ceil(n / gp): ragged for most n here, empty ones at n = 7; slices on either side of the 8-bit / 16-bit window rule)."""
    import msm_amd
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    asz = C.affine_bytes(group)
    bases, scalars = gu.random_case(curve, group, f"gpu:hybrid:{curve}:{group}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    for world in (2, 3, 4, 6, 8):
        for gw in range(1, world + 1):
            if world % gw:
                continue
            parts = _hybrid_parts(C, group, bases, scalars, n, world // gw, gw, asz)
            assert msm_amd.combine(curve, group, parts) == exp, (world, gw)


def test_planned_split_is_a_factorisation_and_covers_the_points(eng):
    """msm_amd_dist_plan for worlds of 1..8 ranks over the sizes of the sweep: gp * gw = world, the slices of the ranks with w = 0
    tile [0, n), ranks of one slice agree on it, and option dist_gw forces the window groups."""
    for logn in (10, 16, 18, 20, 22, 24, 26):
        n = (1 << logn) + 3
        for world in range(1, 9):
            covered = 0
            for rank in range(world):
                p = eng.dist_plan("bls12381", 1, 32, n, world, rank)
                assert p["gp"] * p["gw"] == world
                from msm_amd import dist as mdist
                assert (p["first"], p["count"]) == mdist.slice_of(n, p["gp"], rank // p["gw"])
                if rank % p["gw"] == 0:
                    assert p["first"] == covered
                    covered += p["count"]
            assert covered == n
    try:
        eng.set_option("dist_gw", 2)
        p = eng.dist_plan("bls12381", 1, 32, 1 << 20, 8, 5)
        assert (p["gp"], p["gw"]) == (4, 2) and p["first"] == 2 * (1 << 18) and p["count"] == 1 << 18
        p = eng.dist_plan("bls12381", 1, 32, 1 << 20, 3, 0)         # 2 does not divide 3: the option is ignored
        assert p["gp"] * p["gw"] == 3
    finally:
        eng.set_option("dist_gw", 0)
    # big inputs are cut along the points (every rank a full-size MSM of its slice: profiles/r03_shard_per_rank_table.txt)
    assert eng.dist_plan("bls12381", 1, 32, 1 << 26, 8, 0)["gp"] == 8


def test_exchanges_follow_submission_order_whatever_the_waits_do(orc):
    """A group of one (all a one-GPU box can hold with RCCL itself; tests/test_gpu_two_ranks.py has two) with four slots: G1, G2, G1, G1
    calls submitted, then waited for in REVERSE order -- first by ONE thread (round 4: the wait for a later ticket performs the
    exchanges of the earlier ones on the way and parks their results; in round 3 that order blocked for ever), then from four threads
    at once.  The exchanges themselves stay in submission order -- the order every rank of a real group shares, which keeps two
    promises of a Node host from meeting each other's all-gather -- and every result is right."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("slots", 4)
        eng.dist_init(0, 1, msm_amd.dist_unique_id())
        assert eng.dist_comm_count() == (1, False)          # ncclCommCount of the engine's own communicator
        cases = []
        for k, (curve, group, n) in enumerate((("bls12381", 1, 3000), ("bls12381", 2, 700), ("bls12381", 1, 120000), ("bls12381", 1, 9))):
            bases, scalars = gu.random_case(curve, group, f"gpu:order:{k}", n)
            O = orc.Oracle(curve, group)
            cases.append((group, bases, scalars, n, O.normalize(O.multiexp_affine(bases, scalars, 32, n))))
        C = eng.curve("bls12381")
        tickets = [C.multiexp_dist_async(g, b, s, 32, n) for g, b, s, n, _ in cases]
        for k in (3, 2, 1, 0):                               # one thread, newest first
            assert tickets[k].wait() == cases[k][4], k
        tickets = [C.multiexp_dist_async(g, b, s, 32, n) for g, b, s, n, _ in cases]
        got = [None] * 4

        def waiter(k):
            got[k] = tickets[k].wait()

        threads = []
        for k in (3, 2, 1, 0):
            t = threading.Thread(target=waiter, args=(k,))
            t.start()
            threads.append(t)
        for t in threads:
            t.join(timeout=120)
            assert not t.is_alive()
        for k, (_, _, _, _, exp) in enumerate(cases):
            assert got[k] == exp, k
        # a ticket is collected once
        with pytest.raises(msm_amd.MsmAmdError, match="unknown or already collected"):
            tickets[0].wait()
    finally:
        eng.close()


def test_a_failed_part_completes_the_exchange_and_the_sequence_goes_on(orc):
    """This rank's part of a call cannot be queued (its slice length contradicts the plan; no free slot): the call still gets a
    ticket and takes part in its exchange with a failure mark -- on a real group the peers return MSM_AMD_EPEER instead of hanging in
    the collective -- the error comes back from the wait, and the calls around it are unaffected."""
    import msm_amd
    curve, group, n = "bn128", 1, 2500
    eng = msm_amd.Engine(0)
    try:
        eng.dist_init(0, 1, msm_amd.dist_unique_id())
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, "gpu:distfail", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        plan = eng.dist_plan(curve, group, 32, n)
        assert plan == {"gp": 1, "gw": 1, "first": 0, "count": n}
        good1 = C.multiexp_dist_local_async(group, bases, scalars, 32, n, n)
        bad = C.multiexp_dist_local_async(group, bases, scalars, 32, n - 1, n)       # accepted: the failure travels with the exchange
        assert good1.wait() == exp
        good2 = C.multiexp_dist_local_async(group, bases, scalars, 32, n, n)
        with pytest.raises(msm_amd.MsmAmdError, match="slice of 2500 points is 2500") as ei:
            bad.wait()
        assert ei.value.code == msm_amd.EINVAL
        assert good2.wait() == exp
        # every slot taken (two by default): the third call is refused by the engine, and says so at its wait
        t = [C.multiexp_dist_async(group, bases, scalars, 32, n) for _ in range(3)]
        assert t[0].wait() == exp and t[1].wait() == exp
        with pytest.raises(msm_amd.MsmAmdError, match="every slot"):
            t[2].wait()
        assert C.multiexp_dist(group, bases, scalars, 32, n) == exp
    finally:
        eng.close()


@pytest.mark.parametrize("c", [20, 21, 16, 9])
def test_digits_of_exactly_half_the_window(eng, orc, c):
    """A window digit equal to 2^(c-1) is the largest magnitude a signed c-bit digit takes and lands in the LAST bucket of its window.
    At c = 21 that is 2^20: packed as magnitude << 12 | rank it left the 32-bit word (entry dropped, or LDS indexed out of bounds);
    the coarse partition now packs the bucket index.  Scalars are built so that many windows hold exactly 2^(c-1), with and without a
    carry from the window below (2^(c-1) - 1 plus carry), next to random ones."""
    import random
    curve, group, n = "bls12381", 1, 6000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, rnd_scalars = gu.random_case(curve, group, f"gpu:half:{c}", n)
    rnd = random.Random(f"half:{c}")
    half = 1 << (c - 1)
    W = 255 // c
    sc = bytearray(rnd_scalars)
    for i in range(0, n, 2):
        v = 0
        for w in range(W):
            pick = rnd.randrange(4)
            d = half if pick == 0 else (half - 1 if pick == 1 else (half + 1 if pick == 2 else rnd.randrange(1 << c)))
            v |= d << (w * c)
        sc[32 * i:32 * i + 32] = (v % (1 << 255)).to_bytes(32, "little")
    scalars = bytes(sc)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    try:
        eng.set_window_bits(c)
        for big in (0, 1):
            eng.set_option("sort_big", big)
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp, (c, big)
        assert eng.geometry()["c"] == c
    finally:
        eng.set_window_bits(0)
        eng.set_option("sort_big", 0)


def test_bin_histogram_grows_between_calls(orc):
    """Deterministic form of the sequence behind round 2's one GPU fault: a fresh context makes dozens of calls with the 33 bins of
    8-bit windows, then its first call with more bins than the histogram holds (a 3-way shard at 16-bit windows, tiny n; then full
    calls at 16 and 20 bits).  Whether the allocator hands the regrown buffer back at the old address or not, the new bins must be
    cleared (the mark is keyed on the buffer's allocation count; host-only twin: tests/host/test_devbuf.hip)."""
    import msm_amd
    curve, group = "bls12381", 1
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        for k in range(40):
            n = 5 + 11 * k
            bases, scalars = gu.random_case(curve, group, f"gpu:regrow:{k}", n)
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == O.normalize(O.multiexp_affine(bases, scalars, 32, n))
            assert eng.geometry()["c"] == 8
        n = 7
        bases, scalars = gu.random_case(curve, group, "gpu:regrow:late", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        eng.set_window_bits(16)
        parts = b"".join(C.multiexp_shard(group, bases, scalars, 32, n, r, 3) for r in range(3))
        assert msm_amd.combine(curve, group, parts) == exp
        for c in (16, 20, 13, 21):
            eng.set_window_bits(c)
            n = 3000
            bases, scalars = gu.random_case(curve, group, f"gpu:regrow:c{c}", n)
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == O.normalize(O.multiexp_affine(bases, scalars, 32, n)), c
    finally:
        eng.close()


# ---- prepared bases (msm_amd_prepare_bases / msm_amd_multiexp_prepared) ------------------------------------------------
@pytest.mark.parametrize("curve", ["bls12381", "bn128"])
@pytest.mark.parametrize("group", [1, 2])
def test_prepared_bases_reproduce_the_golden_records(eng, orc, curve, group):
    """Every golden MSM record of the reference WASM (random sizes, n = 0 aside; all-zero / 1 / r-1 / r / 2^256-1 scalars, duplicate
    points, P and -P, affine infinity among the inputs, all points equal, boundary scalars, scalar sizes 1..48, Jacobian bases with
    Z != 1, Z = 1, Z = 0) through a table prepared for the record's scalar width: the same normalised bytes as the reference."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    zero = O.zero()
    seen = 0
    for rec in gu.golden(curve)["groups"][f"g{group}"]["msm"]:
        if rec["n"] == 0:
            continue
        bases, scalars = gu.msm_inputs(curve, group, rec)
        ss = rec["scalar_size"]
        pb = C.prepare_bases(group, bases, rec["n"], scalar_bits=min(8 * ss, 512), jacobian_bases=rec["fn"] != "multiexpAffine")
        try:
            info = pb.info()
            assert info["n"] == rec["n"] and info["windows"] * (info["c"] - 1) + info["signed_windows"] >= min(8 * ss, 512)
            assert pb.multiexp(scalars, ss) == gu.canon(curve, group, bytes.fromhex(rec["out_norm"]), zero), rec["id"]
            seen += 1
        finally:
            pb.release()
    assert seen >= 20


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 20000), ("bn128", 1, 100000), ("bls12381", 2, 3000), ("bn128", 2, 2500), ("bls12381", 1, 3)])
def test_prepared_bases_match_oracle_for_every_window_layout(eng, orc, curve, group, n):
    """Seeded inputs, expected value from the oracle; the table is built for the default scalar width (the group order's bit length:
    scalars are below r) with the window size the engine picks and with forced ones -- 3 bits (all windows unsigned 2-bit or signed
    3-bit), 8 / 9 (one coarse bin), 13, 16, 17 (two-level partition), 20 / 21 (11 / 12 fine bits, 8-byte partition entries) -- then
    prefixes of the points, device-resident scalars, and two calls in flight."""
    import msm_amd
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, f"gpu:prep:{curve}:{group}:{n}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    asz = C.affine_bytes(group)
    try:
        for c in (0, 3, 8, 9, 13, 16, 17, 20, 21):
            eng.set_window_bits(c)
            pb = C.prepare_bases(group, bases, n)
            try:
                info = pb.info()
                if c:
                    assert info["c"] == c
                assert msm_amd.plan_prepared(curve, group, n, 0, c) == info      # the plan made without a GPU is the table that is built
                bits = 255 if curve == "bls12381" else 254
                assert info["scalar_bits"] == bits and info["windows"] * (info["c"] - 1) + info["signed_windows"] >= bits
                assert info["signed_windows"] < info["windows"] and info["table_bytes"] == info["windows"] * ((n + 8191) // 8192 * 8192) * asz
                eng.set_window_bits(0)                       # the call takes its layout from the table, not from the option
                assert pb.multiexp(scalars) == exp, c
                g = eng.geometry()
                assert (g["c"], g["windows"], g["windows_owned"]) == (info["c"], info["windows"], 1)
                if c in (0, 16):
                    m = max(1, n // 3)
                    assert pb.multiexp(scalars[:32 * m], 32, m) == O.normalize(O.multiexp_affine(bases[:asz * m], scalars[:32 * m], 32, m))
                    d_s = eng.alloc(len(scalars)).upload(scalars)
                    t1, t2 = pb.multiexp_async(d_s), pb.multiexp_async(scalars)
                    with pytest.raises(msm_amd.MsmAmdError, match="in flight"):
                        pb.release()
                    assert t1.wait() == exp and t2.wait() == exp
                    d_s.free()
            finally:
                pb.release()
    finally:
        eng.set_window_bits(0)


def test_prepared_bases_edge_inputs_and_wide_scalars(eng, orc):
    """Exceptional inputs on a prepared table -- affine infinity among the bases (its table column stays infinity), repeated points with
    equal digits (the doubling branch inside a bucket that now collects ALL windows), P and -P, scalars 0 / 1 / r - 1 / 2^255 - 1, and the
    carry chain of signed windows running into the unsigned top window -- and the width check: a scalar with a bit beyond the
    prepared width fails loudly (the same inputs succeed on a table prepared for 256 bits, 2^256 - 1 included)."""
    import msm_amd
    curve, group, n = "bls12381", 1, 4000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    asz = 96
    bases, scalars = gu.random_case(curve, group, "gpu:prep:edge", n)
    bases = bytearray(bases)
    bases[asz * 5:asz * 6] = bytes(asz)                                   # affine infinity
    for i in range(200, 300):                                             # one point a hundred times
        bases[asz * i:asz * (i + 1)] = bases[asz * 200:asz * 201]
    y = int.from_bytes(bases[asz * 10 + 48:asz * 11], "little")          # -P = (x, q - y), also in Montgomery form
    bases[asz * 11:asz * 12] = bases[asz * 10:asz * 10 + 48] + (gu.Q_MOD[curve] - y).to_bytes(48, "little")   # P and -P with equal scalars
    bases = bytes(bases)
    r = gu.R_ORDER[curve]
    sc = bytearray(scalars)
    special = [0, 1, r - 1, (1 << 255) - 1, (1 << 254), int("1" + "0" * 19, 2) << 100, ((1 << 255) - 1) ^ (1 << 19)]
    for k, v in enumerate(special):
        sc[32 * (400 + k):32 * (401 + k)] = v.to_bytes(32, "little")
    for i in range(200, 300):
        sc[32 * i:32 * (i + 1)] = sc[32 * 200:32 * 201]                   # equal scalars on the equal points: P + P in every bucket they meet in
    sc[32 * 11:32 * 12] = sc[32 * 10:32 * 11]
    scalars = bytes(sc)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    pb = C.prepare_bases(group, bases, n)                                 # 255 bits
    try:
        assert pb.multiexp(scalars) == gu.canon(curve, group, exp, O.zero())
        wide = bytearray(scalars)
        wide[32 * 77 + 31] |= 0x80                                        # bit 255 of one scalar
        with pytest.raises(msm_amd.MsmAmdError, match="prepared for") as ei:
            pb.multiexp(bytes(wide))
        assert ei.value.code == msm_amd.EINVAL
        assert pb.multiexp(scalars) == gu.canon(curve, group, exp, O.zero())      # the handle and the context stay usable
        # 48-byte scalars whose upper 16 bytes are zero are served; a bit up there is not
        s48 = b"".join(scalars[32 * i:32 * i + 32] + bytes(16) for i in range(n))
        assert pb.multiexp(s48, 48) == gu.canon(curve, group, exp, O.zero())
        w48 = bytearray(s48)
        w48[48 * 9 + 40] = 1
        with pytest.raises(msm_amd.MsmAmdError, match="prepared for"):
            pb.multiexp(bytes(w48), 48)
    finally:
        pb.release()
    pb = C.prepare_bases(group, bases, n, scalar_bits=256)
    try:
        wide = bytearray(scalars)
        wide[32 * 77 + 31] |= 0x80
        wide[32 * 78:32 * 79] = b"\xff" * 32                              # 2^256 - 1
        expw = O.normalize(O.multiexp_affine(bases, bytes(wide), 32, n))
        assert pb.multiexp(bytes(wide)) == gu.canon(curve, group, expw, O.zero())
    finally:
        pb.release()
    # argument checks
    with pytest.raises(msm_amd.MsmAmdError):
        C.prepare_bases(group, bases, 0)
    pb = C.prepare_bases(group, bases, 100)
    try:
        with pytest.raises(msm_amd.MsmAmdError, match="exceeds"):
            pb.multiexp(scalars[:32 * 101], 32, 101)
    finally:
        pb.release()


def test_prepared_bases_at_the_headline_size(eng, orc):
    """n = 2^20 BLS12-381 G1 (the headline workload) on a prepared table: 13 windows of 20 / 19 bits over 2^19 buckets, the table
    built on the GPU from resident bases; equals the general path on the same resident inputs; an oracle spot slice of the same
    points through a prefix call."""
    import torch
    curve, group, n = "bls12381", 1, 1 << 20
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    pts = C.gen_points(group, 0xBA5E5, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    sc[:, 31] &= 0x3F
    torch.cuda.synchronize()
    pb = C.prepare_bases(group, pts, n)
    try:
        info = pb.info()
        assert (info["c"], info["windows"], info["signed_windows"], info["buckets"]) == (20, 13, 8, 1 << 19)
        full = C._multiexp(group, 0, pts, sc, 32, n)
        assert pb.multiexp(sc) == full
        assert eng.geometry()["entries"] > 12 * n
        m = 3000
        h_pts, h_sc = pts.download(m * 96), sc[:m].cpu().numpy().tobytes()
        assert pb.multiexp(sc, 32, m) == O.normalize(O.multiexp_affine(h_pts, h_sc, 32, m))
    finally:
        pb.release()
        pts.free()


def test_prepared_bases_at_four_million_points(eng, orc):
    """n = 2^22 + 4097 on a prepared table (13 windows over 2^19 buckets, 416+ index slices of the 13 x n table, spans of 200 000
    entries cut into chunks): the same point as the general path on the same resident inputs -- one call at a time, two in flight,
    and with either chunk size of the chunk-parallel fine sort -- and linear in the scalars: MSM(s) + MSM(t) = MSM(s + t) for
    scalars whose sum does not leave 255 bits."""
    import torch
    import msm_amd
    curve, group, n = "bls12381", 1, (1 << 22) + 4097
    C = eng.curve(curve)
    pts = C.gen_points(group, 0xF0E4, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4100)
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    t = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    s[:, 31] &= 0x3F
    t[:, 31] &= 0x3F
    # s + t as 256-bit little-endian integers (below 2^255): byte-wise with carry, on the GPU
    acc = s.to(torch.int32) + t.to(torch.int32)
    for k in range(31):
        acc[:, k + 1] += acc[:, k] >> 8
    st = (acc & 0xFF).to(torch.uint8).contiguous()
    torch.cuda.synchronize()
    pb = C.prepare_bases(group, pts, n)
    try:
        info = pb.info()
        assert (info["c"], info["windows"]) == (20, 13)
        full = C._multiexp(group, 0, pts, s, 32, n)
        assert pb.multiexp(s) == full
        for mode in (1, 2):
            eng.set_option("chunk_mode", mode)
            a, b = pb.multiexp_async(s), pb.multiexp_async(t)
            rs, rt = a.wait(), b.wait()
            assert rs == full, mode
        eng.set_option("chunk_mode", 0)
        assert msm_amd.combine(curve, group, rs + rt) == pb.multiexp(st)
        assert pb.multiexp(st) == C._multiexp(group, 0, pts, st, 32, n)
    finally:
        eng.set_option("chunk_mode", 0)
        pb.release()
        pts.free()


def test_index_slices_of_the_partition(eng, orc):
    """4-byte partition entries at every size: indices wider than the entry's index field put their high bits into the bin (S sub-bins
    per (window, coarse bin), csrc/sort_kernels.hpp SortGeom).  With 21-bit windows the field is 19 bits, with 20-bit ones 20: n = 700 001
    points are 2 slices there (the BASELINE sizes reach the same paths at 16-bit windows from n = 2^23: test_gpu_configs).  Every form of
    the fine sort -- chunk-parallel kernels, one workgroup per span chunk by chunk, short spans held in registers (the dense low bins of
    a narrow top window aside, spans at this size are short at c = 21) -- must reproduce the result of the plain 16-bit schedule, which
    the oracle pins on a slice of the same inputs."""
    import torch
    curve, group, n = "bls12381", 1, 700001
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    pts = C.gen_points(group, 0x51CE5, n)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4242)
    sc = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=gen)
    sc[:, 31] &= 0x7F
    torch.cuda.synchronize()
    try:
        m = 2500
        assert C._multiexp(group, 0, pts, sc, 32, m) == O.normalize(O.multiexp_affine(pts.download(m * 96), sc[:m].cpu().numpy().tobytes(), 32, m))
        ref = C._multiexp(group, 0, pts, sc, 32, n)
        assert eng.geometry()["c"] == 16
        for c in (21, 20, 18):
            eng.set_window_bits(c)
            for big in (0, 1, 2):
                eng.set_option("sort_big", big)
                assert C._multiexp(group, 0, pts, sc, 32, n) == ref, (c, big)
            # a window shard and two calls in flight on the sliced partition
            eng.set_option("sort_big", 0)
            import msm_amd
            parts = b"".join(C.multiexp_shard(group, pts, sc, 32, n, r, 3) for r in range(3))
            assert msm_amd.combine(curve, group, parts) == ref, c
            t1, t2 = C.multiexp_async(group, pts, sc, 32, n), C.multiexp_async(group, pts, sc, 32, n)
            assert t1.wait() == ref and t2.wait() == ref
    finally:
        eng.set_window_bits(0)
        eng.set_option("sort_big", 0)
        pts.free()


def test_context_destruction_frees_unreleased_tables(orc):
    """A handle the caller never released: msm_amd_destroy frees its table (the handle is invalid afterwards); a fresh context works."""
    import msm_amd
    curve, group, n = "bn128", 1, 3000
    bases, scalars = gu.random_case(curve, group, "gpu:prep:leak", n)
    O = orc.Oracle(curve, group)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    for _ in range(2):
        eng = msm_amd.Engine(0)
        C = eng.curve(curve)
        pb = C.prepare_bases(group, bases, n)
        assert pb.multiexp(scalars) == exp
        pb.handle = None          # forget it without releasing
        eng.close()


def test_glv_schedule_right_after_skewed_calls(orc):
    """Scalars that use all 256 bits switch 16-bit windows to the even layout over the declared width (1 signed + 16 unsigned windows: no
    carry window; the hint that does it is set by entries in the overflow / carry window).  The GLV schedule keeps its own eight signed 16-bit windows over the 128-bit halves whatever the hint says (a first
    version left it with the even layout's signed-window count: wrong sums, caught by the order of the tests in a shared context)."""
    import msm_amd
    curve, group = "bls12381", 1
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        n = 20000
        bases, _ = gu.random_case(curve, group, "gpu:glvskew", n)
        k = ((1 << 256) - 12345).to_bytes(32, "little")
        C.g1m_multiexpAffine(bases, k * n, 32, n)                       # all scalars equal and 256 bits wide: sets the hint
        full = b"".join((((1 << 255) | (i * 0x9E3779B97F4A7C15)) % (1 << 256)).to_bytes(32, "little") for i in range(n))
        exp = O.normalize(O.multiexp_affine(bases, full, 32, n))
        assert C.g1m_multiexpAffine(bases, full, 32, n) == exp          # 256-bit scalars on the even layout chosen under the hint
        assert (eng.geometry()["c"], eng.geometry()["windows"]) == (8, 33)   # 25 signed 8-bit + 8 unsigned 7-bit windows tile the 256 bits
        eng.set_option("glv", 2)
        m = 4500
        b2, s2 = gu.random_case(curve, group, "gpu:glvskew:2", m)
        assert C.g1m_multiexpAffine(b2, s2, 32, m) == O.normalize(O.multiexp_affine(b2, s2, 32, m))
        assert eng.geometry()["windows"] == 8
        eng.set_option("glv", 0)
        eng.set_window_bits(16)
        assert C.g1m_multiexpAffine(b2, s2, 32, m) == O.normalize(O.multiexp_affine(b2, s2, 32, m))
        assert eng.geometry()["windows"] == 17                          # hint still on: 1 signed + 16 unsigned windows
    finally:
        eng.close()


@pytest.mark.parametrize("curve,order_bits", [("bn128", 254), ("bls12381", 255)])
def test_windows_tile_the_group_orders_width_until_wider_scalars_show_up(orc, curve, order_bits):
    """Where the window size does not divide 256 (14-bit windows here) the windows tile the EXPECTED width of 32-byte scalars -- the
    group order's: 254 bits for BN254, 255 for BLS12-381 -- and one more window takes the bits above, empty for reduced scalars.
    Scalars that do use those bits are still summed correctly (the reference reduces nothing: any 256-bit integer is a valid scalar);
    the entries they put into the overflow window switch the following calls to windows that tile all 256 bits, and sixteen calls
    later the engine tries the narrow layout again."""
    import msm_amd
    group, n = 1, 40000
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, reduced = gu.random_case(curve, group, f"gpu:expected:{curve}", n)           # scalars below r
        exp = O.normalize(O.multiexp_affine(bases, reduced, 32, n))
        c, W_narrow = 14, (order_bits + 14) // 14 + 1
        W_wide = (256 + 14) // 14
        for _ in range(2):
            assert C.g1m_multiexpAffine(bases, reduced, 32, n) == exp
            g = eng.geometry()
            assert (g["c"], g["windows"]) == (c, W_narrow)
            assert g["entries"] <= n * (W_narrow - 1)                                        # nothing in the overflow window
        wide = b"".join(((0xC0 << 248) | int.from_bytes(reduced[32 * i:32 * i + 31], "little")).to_bytes(32, "little") for i in range(n))   # bits 254 and 255 set
        expw = O.normalize(O.multiexp_affine(bases, wide, 32, n))
        assert C.g1m_multiexpAffine(bases, wide, 32, n) == expw                              # narrow layout, overflow window in use
        assert eng.geometry()["windows"] == W_narrow
        assert C.g1m_multiexpAffine(bases, wide, 32, n) == expw                              # the hint: windows over all 256 bits
        assert eng.geometry()["windows"] == W_wide
        seen = []
        for _ in range(18):                                                                  # reduced scalars again: the hint decays
            assert C.g1m_multiexpAffine(bases, reduced, 32, n) == exp
            seen.append(eng.geometry()["windows"])
        assert seen[0] == W_wide and seen[-1] == W_narrow and seen == sorted(seen), seen
        # a sharded call plans without the hint (every rank must choose the same layout from the arguments alone)
        C.g1m_multiexpAffine(bases, wide, 32, n)
        C.g1m_multiexpAffine(bases, wide, 32, n)
        parts = b"".join(C.multiexp_shard(group, bases, wide, 32, n, r, 2) for r in range(2))
        assert msm_amd.combine(curve, group, parts) == expw
    finally:
        eng.close()


def test_prepared_tables_inside_a_group(orc):
    """msm_amd_multiexp_dist_prepared: every rank holds the prepared table of its own slice and its part of a call is a whole prepared
    MSM of that slice (a group of one here: the slice is everything).  Prepared and ordinary calls of a group share the sequence of
    exchanges."""
    import msm_amd
    curve, group, n = "bls12381", 1, 70000
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("slots", 3)
        eng.dist_init(0, 1, msm_amd.dist_unique_id())
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, "gpu:prepdist", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        pb = C.prepare_bases(group, bases, n)
        try:
            assert pb.multiexp_dist(scalars) == exp
            t = [pb.multiexp_dist_async(scalars), C.multiexp_dist_async(group, bases, scalars, 32, n), pb.multiexp_dist_async(scalars, 32, 1000)]
            e1000 = O.normalize(O.multiexp_affine(bases[:96 * 1000], scalars[:32 * 1000], 32, 1000))
            assert t[2].wait.__self__ is t[2]
            got = [None] * 3
            for k in (0, 1, 2):
                got[k] = t[k].wait()
            assert got == [exp, exp, e1000]
            wide = bytearray(scalars)
            wide[31] |= 0x80                                      # a scalar wider than the table: this rank's error, the sequence goes on
            with pytest.raises(msm_amd.MsmAmdError, match="prepared for"):
                pb.multiexp_dist(bytes(wide))
            assert pb.multiexp_dist(scalars) == exp
        finally:
            pb.release()
    finally:
        eng.close()


def test_table_beyond_hbm_is_refused_with_enomem(orc):
    """A table that does not fit the GPU's memory (2^27 BLS12-381 G2 points: 13 windows x 25.8 GB = 335 GB; the plan itself is valid,
    1.74e9 table indices) fails in msm_amd_prepare_bases with MSM_AMD_ENOMEM -- the allocation is the first thing it does, the bases
    are never read -- and leaves the context usable."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve("bls12381")
        n = 1 << 27
        plan = msm_amd.plan_prepared("bls12381", 2, n)
        assert plan["table_bytes"] > 300e9 and plan["windows"] * n < 1 << 31
        d = eng.alloc(n * 192)                       # (uninitialised: never read)
        try:
            with pytest.raises(msm_amd.MsmAmdError) as ei:
                C.prepare_bases(2, d, n)
            assert ei.value.code == msm_amd.ENOMEM, ei.value
        finally:
            d.free()
        O = orc.Oracle("bls12381", 1)
        bases, scalars = gu.random_case("bls12381", 1, "gpu:after-enomem", 300)
        assert C.g1m_multiexpAffine(bases, scalars, 32, 300) == O.normalize(O.multiexp_affine(bases, scalars, 32, 300))
        pb = C.prepare_bases(1, bases, 300)
        assert pb.multiexp(scalars) == O.normalize(O.multiexp_affine(bases, scalars, 32, 300))
        pb.release()
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group,n,opts", [("bls12381", 1, 40000, {}), ("bn128", 1, 40000, {}), ("bls12381", 2, 6000, {"split2": 0}),
                                                ("bls12381", 1, 9000, {"split2": 0, "window_bits": 16}), ("bn128", 2, 5000, {"split2": 0, "window_bits": 12})])
def test_witness_like_scalars_and_their_hot_buckets(orc, curve, group, n, opts):
    """Scalars as a prover's witness has them: 55 % ones, 10 % twos, 10 % zeros, 10 % below 2^16, the rest uniform -- ONE bucket of
    window 0 holds more than half of all points.  Such a bucket is hot (csrc/engine_common.hpp): short segments, its own lane stops
    early, its segment sums are folded by 32 workgroups and then one.  Against the oracle, against the same call with the hot path
    switched off (option "hot" = 0: the ordinary segment path), on a prepared table, and with 64-bit values whose carry makes a hot
    bucket ABOVE their top window."""
    import msm_amd
    import random
    eng = msm_amd.Engine(0)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, uniform = gu.random_case(curve, group, f"gpu:witness:{curve}:{group}:{n}", n)
        rnd = random.Random(n)
        sc = bytearray(uniform)
        for i in range(n):
            r = rnd.random()
            if r < 0.55:
                sc[32 * i:32 * i + 32] = (1).to_bytes(32, "little")
            elif r < 0.65:
                sc[32 * i:32 * i + 32] = (2).to_bytes(32, "little")
            elif r < 0.75:
                sc[32 * i:32 * i + 32] = bytes(32)
            elif r < 0.85:
                sc[32 * i + 2:32 * i + 32] = bytes(30)
        sc = bytes(sc)
        exp = O.normalize(O.multiexp_affine(bases, sc, 32, n))
        f = C.g1m_multiexpAffine if group == 1 else C.g2m_multiexpAffine
        for rep in range(3):                                    # (from the second call on the skew hint shapes the launch grids)
            assert f(bases, sc, 32, n) == exp, rep
        eng.set_option("hot", 0)
        assert f(bases, sc, 32, n) == exp
        eng.set_option("hot", 1)
        pb = C.prepare_bases(group, bases, n)
        try:
            assert pb.multiexp(sc) == exp
        finally:
            pb.release()
        m = n // 2
        wide = b"".join(uniform[32 * i:32 * i + 8] + bytes(24) for i in range(m))     # 64-bit values: half of them carry out of window 3
        assert f(bases, wide, 32, m) == O.normalize(O.multiexp_affine(bases, wide, 32, m))
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group", [("bls12381", 1), ("bn128", 1), ("bn128", 2)])
def test_a_bucket_with_most_of_the_points_folded_by_many_workgroups(orc, curve, group):
    """n = 100 000 with 80 % ones: the bucket of digit 1 holds 80 000 entries -- 5 000 segments of 16, enough (HOT_MIN_SEGS = 4096,
    csrc/engine_common.hpp) for the grouped fold: 32 workgroups sum a stripe of the segment results each, one wave sums the 32 partial
    sums.  Against the oracle, against the ordinary path (option "hot" = 0), one call and two in flight."""
    import msm_amd
    import random
    n = 100000
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, uniform = gu.random_case(curve, group, f"gpu:hotfold:{curve}:{group}", n)
        rnd = random.Random(7)
        one = (1).to_bytes(32, "little")
        sc = b"".join(one if rnd.random() < 0.8 else uniform[32 * i:32 * i + 32] for i in range(n))
        exp = O.normalize(O.multiexp_affine(bases, sc, 32, n))
        for rep in range(3):
            assert C._multiexp(group, 0, bases, sc, 32, n) == exp, rep
        t1, t2 = C.multiexp_async(group, bases, sc, 32, n), C.multiexp_async(group, bases, sc, 32, n)
        assert t1.wait() == exp and t2.wait() == exp
        eng.set_option("hot", 0)
        assert C._multiexp(group, 0, bases, sc, 32, n) == exp
    finally:
        eng.close()


def test_one_prepared_table_shared_by_threads(eng, orc):
    """Six threads call one prepared table and the plain entry point on the same context at once (the libuv workers behind unresolved
    promises of a Node prover): every caller gets its own result; the table cannot be released while a call on it is in flight and
    can right after the last one."""
    import msm_amd
    curve, group, n = "bls12381", 1, 4000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, _ = gu.random_case(curve, group, "gpu:prepthreads", n)
    jobs = []
    for t in range(5):
        _, sc = gu.random_case(curve, group, f"gpu:prepthreads:{t}", n)
        jobs.append((sc, O.normalize(O.multiexp_affine(bases, sc, 32, n))))
    pb = C.prepare_bases(group, bases, n)
    errors, refused = [], []

    def worker(k):
        try:
            for rep in range(5):
                sc, exp = jobs[(k + rep) % len(jobs)]
                got = pb.multiexp(sc) if (k + rep) % 3 else C.g1m_multiexpAffine(bases, sc, 32, n)
                if got != exp:
                    errors.append((k, rep))
                if k == 0 and rep == 2:
                    t = pb.multiexp_async(sc)
                    try:
                        pb.release()
                        errors.append("released with a call in flight")
                    except msm_amd.MsmAmdError as e:
                        refused.append(e.code)
                    if t.wait() != exp:
                        errors.append((k, "async"))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    try:
        assert not errors, errors
        assert refused == [msm_amd.EINVAL]
    finally:
        pb.release()


def _seeds(default):
    """the committed seed, plus any in MSM_TEST_EXTRA_SEEDS (comma-separated; soak runs: tools/runs/r03_soak.sh)"""
    extra = [int(x) for x in os.environ.get("MSM_TEST_EXTRA_SEEDS", "").split(",") if x.strip()]
    return [default] + extra


@pytest.mark.parametrize("seed", _seeds(20261004))
def test_randomised_round3_configurations(eng, orc, seed):
    """60 seeded random combinations of what round 3 added: windows of even width on / off, any window size, scalar sizes whose bit
    count the window does or does not divide, prepared tables with a random declared width and forced window size, hybrid splits of
    2 .. 8 ranks summed on this GPU, the forms of the fine sort and of its chunks -- each against the oracle."""
    import random
    import msm_amd
    from msm_amd import dist as mdist
    rnd = random.Random(seed)
    try:
        for it in range(60):
            curve = rnd.choice(["bls12381", "bn128"])
            group = 1 if rnd.random() < 0.75 else 2
            n = rnd.choice([1, 3, 17, 100, 513, 2049, 6000, 20000] if group == 1 else [1, 9, 130, 700])
            ss = rnd.choice([32, 32, 32, 31, 17, 8, 2, 33, 64])
            C = eng.curve(curve)
            O = orc.Oracle(curve, group)
            asz = C.affine_bytes(group)
            bases, _ = gu.random_case(curve, group, f"gpu:rand3:{it}", n)
            raw = gu.stream(f"gpu:rand3:{it}:s", n * ss)
            # now and then every scalar with all its bits set in the top byte (full-width values) or only a few low bytes in use
            kind = rnd.choice(["uniform", "uniform", "full", "short"])
            sc = bytearray(raw)
            if kind == "full":
                for i in range(n):
                    sc[i * ss + ss - 1] |= 0x80
            elif kind == "short":
                for i in range(n):
                    sc[i * ss + max(1, ss // 2):(i + 1) * ss] = bytes(ss - max(1, ss // 2))
            scalars = bytes(sc)
            c = rnd.choice([0, 0, 3, 5, 9, 12, 13, 14, 15, 17, 19, 20, 21])
            eng.set_window_bits(c)
            eng.set_option("even_windows", rnd.choice([1, 1, 1, 0]))
            eng.set_option("sort_big", rnd.choice([0, 0, 1, 2]))
            eng.set_option("chunk_mode", rnd.choice([0, 1, 2]))
            exp = gu.canon(curve, group, O.normalize(O.multiexp_affine(bases, scalars, ss, n)), O.zero())
            mode = rnd.choice(["full", "prepared", "prepared", "hybrid"])
            tag = (it, curve, group, n, ss, kind, c, mode)
            if mode == "full":
                got = C._multiexp(group, 0, bases, scalars, ss, n)
            elif mode == "prepared":
                bits = rnd.choice([8 * ss, 8 * ss, min(8 * ss + 5, 512)])
                pb = C.prepare_bases(group, bases, n, scalar_bits=bits)
                try:
                    m = n if rnd.random() < 0.7 else max(1, n // 2)
                    got = pb.multiexp(scalars[:m * ss], ss, m)
                    if m != n:
                        exp = gu.canon(curve, group, O.normalize(O.multiexp_affine(bases[:m * asz], scalars[:m * ss], ss, m)), O.zero())
                finally:
                    pb.release()
            else:
                world = rnd.choice([2, 3, 4, 6, 8])
                gw = rnd.choice([w for w in range(1, world + 1) if world % w == 0])
                parts = []
                for r in range(world):
                    first, count = mdist.slice_of(n, world // gw, r // gw)
                    parts.append(C.multiexp_shard(group, bases[first * asz:(first + count) * asz], scalars[first * ss:(first + count) * ss], ss, count, r % gw, gw))
                got = msm_amd.combine(curve, group, b"".join(parts))
            assert got == exp, (tag, eng.geometry())
    finally:
        eng.set_window_bits(0)
        eng.set_option("even_windows", 1)
        eng.set_option("sort_big", 0)
        eng.set_option("chunk_mode", 0)
