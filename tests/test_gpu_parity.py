"""GPU parity tests (run on the MI355X box with -m gpu): the HIP engine, called through the C ABI,
against (a) the committed golden vectors produced by the reference WASM, (b) the CPU oracle on seeded
inputs, (c) size-independent properties at the BASELINE sizes.  Integer work: the bar is bit-exact
equality of the normalised Jacobian bytes."""
import os
import sys

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu

CURVES = ["bls12381", "bn128"]
GROUPS = [1, 2]


@pytest.fixture(scope="module")
def eng():
    import msm_amd
    e = msm_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc(oracle_lib):
    return oracle_lib


def _fn(C, group, affine, chunk=False):
    name = f"g{group}m_multiexp" + ("Affine" if affine else "") + ("_chunk" if chunk else "")
    return getattr(C, name)


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("group", GROUPS)
def test_golden_vectors(eng, orc, curve, group):
    """Every golden MSM record (random, edge classes, scalar sizes, Jacobian bases) incl. _chunk outputs."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    zero = O.zero()
    for rec in gu.golden(curve)["groups"][f"g{group}"]["msm"]:
        bases, scalars = gu.msm_inputs(curve, group, rec)
        affine = rec["fn"] == "multiexpAffine"
        out = _fn(C, group, affine)(bases, scalars, rec["scalar_size"], rec["n"])
        assert out == gu.canon(curve, group, bytes.fromhex(rec["out_norm"]), zero), rec["id"]
        for ch in rec.get("chunks", []):
            o = _fn(C, group, affine, True)(bases, scalars, rec["scalar_size"], rec["n"], ch["start_bit"], ch["chunk_size"])
            assert o == gu.canon(curve, group, bytes.fromhex(ch["out_norm"]), zero), (rec["id"], ch["start_bit"], ch["chunk_size"])


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("group", GROUPS)
@pytest.mark.parametrize("c", [0, 1, 2, 5, 9, 13])
def test_window_size_independence(eng, orc, curve, group, c):
    """The window schedule is not part of the contract: every window size gives the reference's answer."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    rec = [r for r in gu.golden(curve)["groups"][f"g{group}"]["msm"] if r["id"] == "rand_n64"][0]
    bases, scalars = gu.msm_inputs(curve, group, rec)
    eng.set_window_bits(c)
    try:
        out = _fn(C, group, True)(bases, scalars, 32, 64)
        assert out == bytes.fromhex(rec["out_norm"])
        assert eng.geometry()["c"] == (c if c else eng.geometry()["c"])
    finally:
        eng.set_window_bits(0)


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 5000), ("bls12381", 1, 16384), ("bn128", 1, 16384),
                                           ("bls12381", 2, 3000), ("bn128", 2, 3000)])
def test_oracle_seeded(eng, orc, curve, group, n):
    """Seeded inputs drawn from the committed pool; expected value from the CPU oracle."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, f"gpu:{curve}:{group}:{n}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    assert _fn(C, group, True)(bases, scalars, 32, n) == exp
    # a few windows through the _chunk entry point
    for start, cs in [(0, 13), (247, 13), (100, 7)]:
        e = O.normalize(O.multiexp_chunk(bases, scalars, 32, n, start, cs))
        assert _fn(C, group, True, True)(bases, scalars, 32, n, start, cs) == gu.canon(curve, group, e, O.zero())


def test_reference_kats(eng, orc):
    import json
    with open(os.path.join(gu.GOLDEN, "kats_sum_of_squares.json")) as f:
        kats = json.load(f)
    for k in kats["cases"]:
        C = eng.curve(k["curve"])
        out = _fn(C, k["group"], True)(bytes.fromhex(k["bases"]), bytes.fromhex(k["scalars"]), k["scalar_size"], k["n"])
        assert out.hex() == k["expect_norm"]
    with open(os.path.join(gu.GOLDEN, "kats_batchaffine.json")) as f:
        k = json.load(f)["multiexp_case1"]
    import msm_amd
    plain = b"".join(int(h, 16).to_bytes(48, "little") for h in k["points_affine_plain"])
    bases = msm_amd.f_to_montgomery("bls12381", plain)
    scalars = b"".join(int(h, 16).to_bytes(32, "little") for h in k["scalars"])
    out = eng.curve("bls12381").g1m_multiexp_multiExp(bases, scalars, 10)    # opt-path signature
    aff = msm_amd.f_from_montgomery("bls12381", msm_amd.to_affine("bls12381", 1, out))
    assert [hex(int.from_bytes(aff[:48], "little")), hex(int.from_bytes(aff[48:], "little"))] == k["expected_affine_plain"]
    # the chunk-level tests of the opt path (batchAffine.js:884-977, :979-1056, :1058-1175) as MSMs, whole and per chunk
    with open(os.path.join(gu.GOLDEN, "kats_batchaffine.json")) as f:
        sched = json.load(f)["scheduled"]
    O = orc.Oracle("bls12381", 1)
    C = eng.curve("bls12381")
    for name, k in sched.items():
        plain = b"".join(int(h, 16).to_bytes(48, "little") for h in k["points_affine_plain"])
        bases = msm_amd.f_to_montgomery("bls12381", plain)
        scalars = b"".join(int(h, 16).to_bytes(32, "little") for h in k["scalars"])
        aff = msm_amd.f_from_montgomery("bls12381", msm_amd.to_affine("bls12381", 1, C.g1m_multiexpAffine(bases, scalars, 32, 10)))
        assert [hex(int.from_bytes(aff[:48], "little")), hex(int.from_bytes(aff[48:], "little"))] == k["expected_affine_plain"], name
        acc = O.zero()
        for c in range(k["num_chunks"] - 1, -1, -1):
            for _ in range(k["chunk_size"]):
                acc = O.double(acc)
            acc = O.add(acc, C.g1m_multiexpAffine_chunk(bases, scalars, 32, 10, c * k["chunk_size"], k["chunk_size"]))
        aff = msm_amd.f_from_montgomery("bls12381", O.to_affine(acc))
        assert [hex(int.from_bytes(aff[:48], "little")), hex(int.from_bytes(aff[48:], "little"))] == k["expected_affine_plain"], name


def test_adversarial_single_bucket(eng, orc):
    """All scalars equal: every pair lands in one bucket per window (maximal imbalance, P+P chains)."""
    curve, group, n = "bls12381", 1, 2048
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, _ = gu.random_case(curve, group, "gpu:adv", n)
    s = gu.scalars_for(curve, "gpu:adv:one", 1)
    scalars = s * n
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp
    # all points equal as well
    bases2 = bases[:96] * n
    exp2 = O.normalize(O.multiexp_affine(bases2, scalars, 32, n))
    assert C.g1m_multiexpAffine(bases2, scalars, 32, n) == exp2


@pytest.mark.parametrize("curve,group", [("bls12381", 1), ("bn128", 1), ("bls12381", 2)])
def test_long_buckets_are_segmented(eng, orc, curve, group):
    """Force the long-bucket path (segment length 4): buckets are cut into segments summed by the tail
    kernels; also the natural trigger -- all scalars equal at a size where one lane per bucket would crawl."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    n = 3000
    bases, scalars = gu.random_case(curve, group, f"gpu:seg:{curve}:{group}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    eng.set_option("segment", 4)
    eng.set_window_bits(6)
    try:
        assert _fn(C, group, True)(bases, scalars, 32, n) == exp
        same = scalars[:32] * n
        exp2 = O.normalize(O.multiexp_affine(bases, same, 32, n))
        assert _fn(C, group, True)(bases, same, 32, n) == exp2
    finally:
        eng.set_option("segment", 0)
        eng.set_window_bits(0)
    # natural trigger at default settings: all scalars equal, n = 2^15 (one bucket per window holds everything)
    if group == 1:
        m = 1 << 15
        b2 = (bases * (m // n + 1))[:m * C.affine_bytes(group)]
        same = scalars[:32] * m
        exp3 = O.normalize(O.multiexp_affine(b2, same, 32, m))
        assert _fn(C, group, True)(b2, same, 32, m) == exp3


def test_async_pipeline_two_in_flight(eng, orc):
    """msm_amd_multiexp_async / msm_amd_wait: two calls in flight, different curves and sizes, results
    collected out of submission order; a third submission is refused until a ticket is collected."""
    import msm_amd
    cases = []
    for curve, group, n in (("bls12381", 1, 4000), ("bn128", 1, 700), ("bls12381", 2, 300), ("bls12381", 1, 0)):
        bases, scalars = gu.random_case(curve, group, f"gpu:async:{curve}:{group}:{n}", n) if n else (b"", b"")
        O = orc.Oracle(curve, group)
        cases.append((curve, group, n, bases, scalars, O.normalize(O.multiexp_affine(bases, scalars, 32, n))))
    t0 = eng.curve(cases[0][0]).multiexp_async(cases[0][1], cases[0][3], cases[0][4], 32, cases[0][2])
    t1 = eng.curve(cases[1][0]).multiexp_async(cases[1][1], cases[1][3], cases[1][4], 32, cases[1][2])
    with pytest.raises(msm_amd.MsmAmdError):
        eng.curve("bls12381").multiexp_async(1, cases[0][3], cases[0][4], 32, cases[0][2])
    with pytest.raises(msm_amd.MsmAmdError):      # a synchronous call cannot wait for tickets only its own caller can collect
        eng.curve("bn128").g1m_multiexpAffine(cases[1][3], cases[1][4], 32, cases[1][2])
    assert t1.wait() == cases[1][5]
    t2 = eng.curve(cases[2][0]).multiexp_async(cases[2][1], cases[2][3], cases[2][4], 32, cases[2][2])
    assert t0.wait() == cases[0][5]
    t3 = eng.curve(cases[3][0]).multiexp_async(cases[3][1], cases[3][3], cases[3][4], 32, cases[3][2])
    assert t2.wait() == cases[2][5]
    assert t3.wait() == cases[3][5]
    with pytest.raises(msm_amd.MsmAmdError):
        t3.wait()                                   # already collected
    # steady-state pipeline on one input: every result identical
    outs = []
    prev = None
    for _ in range(6):
        cur = eng.curve(cases[0][0]).multiexp_async(cases[0][1], cases[0][3], cases[0][4], 32, cases[0][2])
        if prev is not None:
            outs.append(prev.wait())
        prev = cur
    outs.append(prev.wait())
    assert all(o == cases[0][5] for o in outs)
    # four slots: four calls in flight, collected in reverse order
    eng.set_option("slots", 4)
    try:
        ts = [eng.curve(c[0]).multiexp_async(c[1], c[3], c[4], 32, c[2]) for c in cases]
        for tk, c in reversed(list(zip(ts, cases))):
            assert tk.wait() == c[5]
    finally:
        eng.set_option("slots", 2)


def _seeds(default):
    """the committed seed, plus any in MSM_TEST_EXTRA_SEEDS (comma-separated; soak runs: tools/runs/r03_soak.sh)"""
    extra = [int(x) for x in os.environ.get("MSM_TEST_EXTRA_SEEDS", "").split(",") if x.strip()]
    return [default] + extra


@pytest.mark.parametrize("seed", _seeds(20261003))
def test_randomised_configurations(eng, orc, seed):
    """40 seeded random combinations of curve, n, scalar size, window bits, segment length, reduction fan-out and
    entry point (full / chunk / shard), each against the oracle."""
    import random
    import msm_amd
    rnd = random.Random(seed)
    try:
        for it in range(40):
            curve = rnd.choice(["bls12381", "bn128"])
            group = 1 if rnd.random() < 0.75 else 2
            n = rnd.choice([1, 2, 7, 33, 100, 257, 1000, 2500] if group == 1 else [1, 5, 64, 300])
            ss = rnd.choice([32, 32, 32, 31, 16, 8, 3, 40, 64])
            C = eng.curve(curve)
            O = orc.Oracle(curve, group)
            bases, _ = gu.random_case(curve, group, f"gpu:rand:{it}", n)
            scalars = gu.stream(f"gpu:rand:{it}:s", n * ss)
            eng.set_window_bits(rnd.choice([0, 0, 1, 3, 4, 7, 11, 16]))
            eng.set_option("segment", rnd.choice([0, 0, 1, 2, 5, 64]))
            eng.set_option("reduce_l", rnd.choice([0, 0, 1, 2, 16]))
            eng.set_option("split", rnd.choice([0, 1]))
            mode = rnd.choice(["full", "full", "chunk", "shard"])
            if os.environ.get("MSM_TEST_TRACE"):
                print("randomised", it, curve, group, n, ss, mode, file=sys.stderr, flush=True)
            if mode == "full":
                exp = O.normalize(O.multiexp_affine(bases, scalars, ss, n))
                got = _fn(C, group, True)(bases, scalars, ss, n)
            elif mode == "chunk":
                start = rnd.randrange(0, ss * 8)
                cs = rnd.randrange(1, 26)
                exp = O.normalize(O.multiexp_chunk(bases, scalars, ss, n, start, cs))
                got = _fn(C, group, True, True)(bases, scalars, ss, n, start, cs)
            else:
                k = rnd.choice([2, 3, 5])
                exp = O.normalize(O.multiexp_affine(bases, scalars, ss, n))
                got = msm_amd.combine(curve, group, b"".join(C.multiexp_shard(group, bases, scalars, ss, n, r, k) for r in range(k)))
            assert got == gu.canon(curve, group, exp, O.zero()), (it, curve, group, n, ss, mode, eng.geometry())
    finally:
        eng.set_window_bits(0)
        eng.set_option("segment", 0)
        eng.set_option("reduce_l", 0)
        eng.set_option("split", 0)


def test_batch_conversions(eng, orc):
    """g?m_batchToAffine / _batchToJacobian / _batchLEMtoU / _batchUtoLEM on the GPU: byte-equal to the reference's
    outputs (tests/golden/golden_conv.json) and to the oracle on a larger seeded batch."""
    import json
    with open(os.path.join(gu.GOLDEN, "golden_conv.json")) as f:
        conv = json.load(f)
    for c in conv["cases"]:
        C = eng.curve(c["curve"])
        g = c["group"]
        jac, aff = bytes.fromhex(c["jacobian_in"]), bytes.fromhex(c["affine_in"])
        assert getattr(C, f"g{g}m_batchToAffine")(jac).hex() == c["batchToAffine"]
        assert getattr(C, f"g{g}m_batchToJacobian")(aff).hex() == c["batchToJacobian"]
        u = getattr(C, f"g{g}m_batchLEMtoU")(aff)
        assert u.hex() == c["batchLEMtoU"]
        assert getattr(C, f"g{g}m_batchUtoLEM")(u).hex() == c["batchUtoLEM_of_U"]
        # larger batch against the oracle (1000 points incl. repeated infinities; not a multiple of the lane batch)
        O = orc.Oracle(c["curve"], g)
        n = 1003
        big_aff = (aff * (n // c["n"] + 1))[:n * O.asz]
        big_jac = (jac * (n // c["n"] + 1))[:n * O.jsz]
        assert getattr(C, f"g{g}m_batchToAffine")(big_jac) == O.batch_to_affine(big_jac)
        assert getattr(C, f"g{g}m_batchToJacobian")(big_aff) == O.batch_to_jacobian(big_aff)
        assert getattr(C, f"g{g}m_batchLEMtoU")(big_aff) == O.batch_lem_to_u(big_aff)
        assert getattr(C, f"g{g}m_batchUtoLEM")(O.batch_lem_to_u(big_aff)) == big_aff
        assert getattr(C, f"g{g}m_batchToAffine")(b"") == b""
        assert C.batch_in_curve_affine(g, aff) == c["inCurveAffine"]
        assert C.batch_in_curve_affine(g, bytes.fromhex(c["swapped_in"])) == c["inCurveAffine_swapped"]
        assert C.batch_in_curve_affine(g, big_aff) == O.batch_in_curve_affine(big_aff)


def test_split_window_groups(eng, orc):
    """Option "split": the windows of a lone call accumulated and reduced as two overlapping groups -- same result."""
    curve, group, n = "bls12381", 1, 6000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, "gpu:split", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    eng.set_option("split", 1)
    eng.set_window_bits(16)
    try:
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp
        assert eng.geometry()["c"] == 16
    finally:
        eng.set_option("split", 0)
        eng.set_window_bits(0)


@pytest.mark.parametrize("group", [1, 2])
@pytest.mark.parametrize("curve", ["bls12381", "bn128"])
def test_several_lanes_per_bucket(eng, orc, curve, group):
    """Launches with few, well-filled buckets give every bucket several lanes (k_accumulate_slices + k_join_slices): same bytes as
    with one lane, with odd and single-entry buckets (3001 entries over 2^7 buckets, 101 over 2^2), with long buckets cut into
    segments, and with equal partial sums meeting at the join of the slices (P + P)."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    A = C.affine_bytes(group)
    try:
        for n, c, seg in [(3001, 8, 0), (101, 3, 0), (1500, 6, 5), (2048, 4, 0)]:   # mean population >= 16: the kernel applies
            bases, scalars = gu.random_case(curve, group, f"gpu:split2:{curve}:{group}:{n}:{c}", n)
            if n == 2048:   # all pairs equal: the slices of a bucket sum to the same point, the join is a doubling
                bases = bases[:A] * n
                scalars = scalars[:32] * n
            exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
            eng.set_window_bits(c)
            eng.set_option("segment", seg)
            eng.set_option("split2", 0)
            one_lane = _fn(C, group, True)(bases, scalars, 32, n)
            assert one_lane == gu.canon(curve, group, exp, O.zero()), (curve, n, c, seg)
            eng.set_option("split2", 1)
            for k in (0, 2, 3, 7, 8, 16, 64):   # 8 and up, powers of two: the tree join
                eng.set_option("splitk", k)
                assert _fn(C, group, True)(bases, scalars, 32, n) == one_lane, (curve, group, n, c, seg, k)
        with pytest.raises(Exception):
            eng.set_option("splitk", 65)
    finally:
        eng.set_option("split2", 1)
        eng.set_option("splitk", 0)
        eng.set_option("segment", 0)
        eng.set_window_bits(0)


def test_batch_apply_key_matches_reference(eng, orc):
    """g?m_batchApplyKey / g?m_batchApplyKeyMixed / frm_batchApplyKey on the GPU == the reference WASM (golden_applykey.json):
    Fr elements byte for byte, group elements as normalised Jacobian points; and a longer batch against the oracle."""
    import json
    import os
    with open(os.path.join(gu.GOLDEN, "golden_applykey.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        C = eng.curve(c["curve"])
        first, inc = bytes.fromhex(c["first"]), bytes.fromhex(c["inc"])
        assert C.frm_batchApplyKey(bytes.fromhex(c["fr_in"]), first, inc).hex() == c["fr_out"]
        for g in (1, 2):
            O = orc.Oracle(c["curve"], g)
            G = c["groups"][f"g{g}"]
            zero = O.zero()
            for key_in, jac, key_out in (("affine_in", False, "mixed_out_norm"), ("jacobian_in", True, "jac_out_norm")):
                got = C.batch_apply_key(g, bytes.fromhex(G[key_in]), first, inc, jac)
                exp = bytes.fromhex(G[key_out])
                exp = b"".join(gu.canon(c["curve"], g, exp[i * O.jsz:(i + 1) * O.jsz], zero) for i in range(G["n"]))
                assert got == exp, (c["curve"], c["variant"], g, key_in)
    # indices beyond the golden batch (the per-lane power inc^i): 300 points against the oracle
    c = cases[0]
    first, inc = bytes.fromhex(c["first"]), bytes.fromhex(c["inc"])
    O = orc.Oracle(c["curve"], 1)
    pts = gu.pool_points(c["curve"], 1, [i % 40 for i in range(300)])
    raw = O.batch_apply_key(pts, first, inc, False)
    exp = b"".join(O.normalize(raw[i * O.jsz:(i + 1) * O.jsz]) for i in range(300))
    assert eng.curve(c["curve"]).g1m_batchApplyKeyMixed(pts, first, inc) == exp


def test_fft_matches_reference(eng, orc):
    """frm_fft / frm_ifft byte for byte and g?m_fft / g?m_ifft as normalised points == the reference WASM (golden_fft.json); a
    larger Fr transform against the oracle and a round trip at 2^16."""
    import json
    import os
    with open(os.path.join(gu.GOLDEN, "golden_fft.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        C = eng.curve(c["curve"])
        inp = bytes.fromhex(c["in"])
        if c["group"] == 0:
            assert C.frm_fft(inp).hex() == c["fft"], (c["curve"], c["n"])
            assert C.frm_ifft(inp).hex() == c["ifft"], (c["curve"], c["n"])
        else:
            O = orc.Oracle(c["curve"], c["group"])
            zero = O.zero()
            for inv, key in ((False, "fft"), (True, "ifft")):
                exp = bytes.fromhex(c[key])
                exp = b"".join(gu.canon(c["curve"], c["group"], exp[i * O.jsz:(i + 1) * O.jsz], zero) for i in range(c["n"]))
                assert C.fft(c["group"], inp, inv) == exp, (c["curve"], c["group"], c["n"], key)
    for curve in CURVES:
        C = eng.curve(curve)
        O = orc.Oracle(curve, 1)
        r = gu.R_ORDER[curve]
        raw = gu.stream(f"gpu:fft:{curve}", 40 * 4096)
        plain = b"".join((int.from_bytes(raw[40 * i:40 * i + 40], "little") % r).to_bytes(32, "little") for i in range(4096))
        assert C.frm_fft(plain) == O.fr_fft(plain)
        assert C.frm_ifft(plain) == O.fr_fft(plain, True)
        big = plain * 16                                                    # 2^16 elements
        assert C.frm_ifft(C.frm_fft(big)) == big
    with pytest.raises(Exception):
        eng.curve("bls12381").frm_fft(bytes(32 * 3))                         # not a power of two


def test_compressed_form_matches_reference(eng, orc):
    """g?m_batchLEMtoC / g?m_batchCtoLEM on the GPU == the reference WASM (golden_compressed.json) and the oracle on a larger batch;
    an x whose x^3 + b is not a square is reported, not looped on."""
    import json
    import os
    with open(os.path.join(gu.GOLDEN, "golden_compressed.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        C = eng.curve(c["curve"])
        O = orc.Oracle(c["curve"], c["group"])
        g = c["group"]
        aff, comp = bytes.fromhex(c["affine_in"]), bytes.fromhex(c["compressed"])
        to_c = getattr(C, f"g{g}m_batchLEMtoC")
        to_lem = getattr(C, f"g{g}m_batchCtoLEM")
        assert to_c(aff) == comp
        assert to_lem(comp) == aff
        big = gu.pool_points(c["curve"], g, list(range(200)))
        assert to_c(big) == O.batch_lem_to_c(big)
        assert to_lem(to_c(big)) == big
    C = eng.curve("bls12381")
    bad = None
    for x in range(1, 40):                                    # find an x with x^3 + 4 a non-residue
        comp = x.to_bytes(48, "big")
        try:
            orc.Oracle("bls12381", 1).batch_c_to_lem(comp)
        except ValueError:
            bad = comp
            break
    assert bad is not None
    with pytest.raises(Exception):
        C.g1m_batchCtoLEM(bad)


def test_in_group_affine_matches_reference(eng, orc):
    """Batched subgroup check on the GPU == the reference's g?m_inGroupAffine flags (golden_ingroup.json) and the oracle's."""
    import json
    import os
    with open(os.path.join(gu.GOLDEN, "golden_ingroup.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        C = eng.curve(c["curve"])
        O = orc.Oracle(c["curve"], c["group"])
        aff = bytes.fromhex(c["affine_in"])
        assert C.batch_in_group_affine(c["group"], aff) == c["inGroupAffine"]
        big = aff * 7 + gu.pool_points(c["curve"], c["group"], list(range(40)))
        assert C.batch_in_group_affine(c["group"], big) == O.batch_in_group_affine(big)
    with pytest.raises(Exception):
        eng.curve("bn128").batch_in_group_affine(1, bytes(64))


def test_glv_path_matches_oracle_and_plain_path(eng, orc):
    """BLS12-381 G1 with whole 32-byte scalars and c = 16 runs through the engine's GLV split (2n virtual points, 8 windows):
    same bytes as the oracle and as the plain 16-window path, including scalars >= r, around r/2, 0, 1 and an affine infinity."""
    curve, group, n = "bls12381", 1, 4500
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    r = gu.R_ORDER[curve]
    bases, scalars = gu.random_case(curve, group, "gpu:glvpath", n)
    u0 = 0xac45a4010001a4020000000100000000
    edge = [0, 1, 2, r - 1, r, r + 1, 2 * r + 7, (1 << 256) - 1, 1 << 255, (r - 1) // 2, (r + 1) // 2, (1 << 128) - 1, 1 << 127,
            u0 // 2, u0, u0 + 1, r - u0, 0x8000, 0x80008000, r - 0x8000, u0 - 0x8000, (0x8000 << 112) | 0x8000]   # digits equal to 2^15 under both signs
    scalars = b"".join(v.to_bytes(32, "little") for v in edge) + scalars[32 * len(edge):]
    bases = bases[:96 * 20] + bytes(96) + bases[96 * 21:]            # affine infinity among the inputs
    bases = bases[:96 * 30] + bases[96 * 31:96 * 32] + bases[96 * 31:]   # a repeated point
    expect = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    try:
        eng.set_option("glv", 2)
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == expect
        assert eng.geometry()["windows"] == 8
        assert C.g1m_multiexp(O.batch_to_jacobian(bases), scalars, 32, n) == expect     # Jacobian bases: converted, then the same path
        eng.set_option("glv", 0)
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == expect
        assert eng.geometry()["windows"] == 33                      # n < 98304: 8-bit windows, 32 + the carry
        eng.set_window_bits(16)
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == expect
        assert eng.geometry()["windows"] == 17
    finally:
        eng.set_window_bits(0)
        eng.set_option("glv", 1)


def test_point_slicing_of_large_synchronous_calls(eng, orc):
    """More points than one pass takes are cut into slices (two in flight) and the partial results added on the host.  The slice size
    is lowered with the "slice" option so that the path runs at test sizes: full calls, a _chunk call, Jacobian bases, G2, and a
    window shard all give the oracle's answer."""
    import msm_amd
    try:
        eng.set_option("slice", 1000)
        eng.set_option("slots", 1)                       # a single slot must not wait on itself
        C = eng.curve("bn128")
        O = orc.Oracle("bn128", 1)
        bases, scalars = gu.random_case("bn128", 1, "gpu:slices:oneslot", 3100)
        assert C.g1m_multiexpAffine(bases, scalars, 32, 3100) == O.normalize(O.multiexp_affine(bases, scalars, 32, 3100))
        eng.set_option("slots", 2)
        for curve, group, n in (("bls12381", 1, 4500), ("bn128", 2, 2300)):
            C = eng.curve(curve)
            O = orc.Oracle(curve, group)
            bases, scalars = gu.random_case(curve, group, f"gpu:slices:{curve}:{group}", n)
            bases = bases[:O.asz * 7] + bytes(O.asz) + bases[O.asz * 8:]
            exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
            assert _fn(C, group, True)(bases, scalars, 32, n) == exp
            assert _fn(C, group, False)(O.batch_to_jacobian(bases), scalars, 32, n) == exp
            chunk = O.normalize(O.multiexp_chunk(bases, scalars, 32, n, 37, 11))
            assert _fn(C, group, True, True)(bases, scalars, 32, n, 37, 11) == chunk
            parts = b"".join(C.multiexp_shard(group, bases, scalars, 32, n, r, 3) for r in range(3))
            assert msm_amd.combine(curve, group, parts) == exp
    finally:
        eng.set_option("slots", 2)
        eng.set_option("slice", 1 << 26)


def test_concurrent_calls_from_threads(eng, orc):
    """Several threads issue synchronous calls on ONE context (what the libuv workers behind unresolved JS promises do): the
    context serialises submission and collection, overlaps the GPU work, and every caller gets its own result."""
    import threading
    curve, group, n = "bls12381", 1, 3000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, _ = gu.random_case(curve, group, "gpu:threads", n)
    jobs = []
    for t in range(6):
        _, sc = gu.random_case(curve, group, f"gpu:threads:{t}", n)
        jobs.append((sc, O.normalize(O.multiexp_affine(bases, sc, 32, n))))
    errors = []

    def worker(k):
        try:
            for rep in range(4):
                sc, exp = jobs[(k + rep) % len(jobs)]
                if C.g1m_multiexpAffine(bases, sc, 32, n) != exp:
                    errors.append((k, rep))
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_resident_bases_host_scalars(eng, orc):
    """MSM_AMD_DEVICE_BASES: the bases stay on the GPU across calls (a prover's fixed SRS), scalars come from the host."""
    curve, group, n = "bn128", 1, 3000
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, s1 = gu.random_case(curve, group, "gpu:resident:1", n)
    _, s2 = gu.random_case(curve, group, "gpu:resident:2", n)
    dev = eng.alloc(len(bases)).upload(bases)
    try:
        for sc in (s1, s2):
            assert C.g1m_multiexpAffine(dev, sc, 32, n) == O.normalize(O.multiexp_affine(bases, sc, 32, n))
    finally:
        dev.free()


def test_error_codes(eng):
    import msm_amd
    C = eng.curve("bls12381")
    with pytest.raises(msm_amd.MsmAmdError):
        C.g1m_multiexpAffine(b"\0" * 96, b"\0" * 65, 65, 1)       # scalar_size > 64
    with pytest.raises(msm_amd.MsmAmdError):
        C.g1m_multiexpAffine_chunk(b"\0" * 96, b"\0" * 32, 32, 1, 0, 26)   # chunk_size > 25
    with pytest.raises(msm_amd.MsmAmdError):
        eng.curve("nope")
    assert C.g1m_multiexpAffine(b"", b"", 32, 0) == msm_amd.combine("bls12381", 1, b"")    # n = 0 -> infinity


@pytest.mark.parametrize("curve,group", [("bls12381", 1), ("bn128", 1), ("bls12381", 2), ("bn128", 2)])
def test_generated_points_are_multiples_of_generator(eng, orc, curve, group):
    """msm_amd_gen_points (synthetic bench input): P_i = k_i * G with k_i from splitmix64."""
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    seed, n = 1234, 300
    buf = C.gen_points(group, seed, n)
    pts = buf.download(n * C.affine_bytes(group))
    buf.free()
    M = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
        return x ^ (x >> 31)
    gen = bytes.fromhex(gu.golden(curve)["groups"][f"g{group}"]["generator_jacobian"])
    asz = C.affine_bytes(group)
    for i in (0, 1, 63, 64, 299):
        k = [sm((seed * 0x100000001B3 + i * 4 + j) & M) for j in range(4)]
        k[3] &= 0x3fffffffffffffff
        kv = k[0] | (k[1] << 64) | (k[2] << 128) | (k[3] << 192)
        exp = O.to_affine(O.times_scalar(gen, kv.to_bytes(32, "little")))
        assert pts[i * asz:(i + 1) * asz] == exp, i


def _rand_scalars(n, nbytes, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, nbytes:] = 0
    return a


@pytest.mark.parametrize("curve,group,logn", [("bls12381", 1, 20), ("bn128", 1, 20), ("bls12381", 2, 18)])
def test_full_size_properties(eng, orc, curve, group, logn):
    """BASELINE-size checks that need no oracle run: linearity in the scalars, slice additivity,
    window (chunk) decomposition, and an oracle spot check on a 2^11 slice of the same inputs."""
    import msm_amd
    import numpy as np
    n = 1 << logn
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    asz = C.affine_bytes(group)
    pts = C.gen_points(group, 77, n)
    s = _rand_scalars(n, 31, 1)
    t = _rand_scalars(n, 31, 2)
    # s + t as 256-bit integers (31-byte operands: no overflow)
    carry = np.zeros(n, dtype=np.uint16)
    u = np.zeros_like(s)
    for b in range(32):
        v = s[:, b].astype(np.uint16) + t[:, b].astype(np.uint16) + carry
        u[:, b] = (v & 0xFF).astype(np.uint8)
        carry = v >> 8
    ds, dt, du = (eng.alloc(n * 32).upload(x.tobytes()) for x in (s, t, u))
    f = _fn(C, group, True)
    Rs, Rt, Ru = f(pts, ds, 32, n), f(pts, dt, 32, n), f(pts, du, 32, n)
    assert msm_amd.combine(curve, group, Rs + Rt) == Ru                    # linearity
    # slice additivity
    half = n // 2
    host_pts = pts.download(n * asz)
    lo = f(host_pts[:half * asz], s[:half].tobytes(), 32, half)
    hi = f(host_pts[half * asz:], s[half:].tobytes(), 32, n - half)
    assert msm_amd.combine(curve, group, lo + hi) == Rs
    # oracle spot check on a slice
    m = 2048
    exp = O.normalize(O.multiexp_affine(host_pts[:m * asz], s[:m].tobytes(), 32, m))
    assert f(host_pts[:m * asz], s[:m].tobytes(), 32, m) == exp
    # window decomposition through the _chunk entry point: sum_w 2^(16 w) chunk_w == full (host Horner via oracle ops)
    fc = _fn(C, group, True, True)
    acc = O.zero()
    for w in range(15, -1, -1):
        for _ in range(16):
            acc = O.double(acc)
        acc = O.add(acc, fc(pts, ds, 32, n, 16 * w, 16))
    assert O.normalize(acc) == Rs
    # the eight window shards of the multi-GPU decomposition add up to the full result at full size (for G1 a rank's two or three
    # windows are few enough buckets for the sliced accumulation)
    parts = b"".join(C.multiexp_shard(group, pts, ds, 32, n, r, 8) for r in range(8))
    assert msm_amd.combine(curve, group, parts) == Rs
    # the GLV schedule (8 windows over 2n half-length scalars) and the plain 16-window schedule agree at full size
    if (curve, group) == ("bls12381", 1):
        try:
            eng.set_option("glv", 2)
            Rg = f(pts, ds, 32, n)
            assert eng.geometry()["windows"] == 8
        finally:
            eng.set_option("glv", 1)
        assert Rg == Rs
    for b in (ds, dt, du, pts):
        b.free()


@pytest.mark.parametrize("curve,group", [("bls12381", 1), ("bn128", 2)])
@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_window_shards_sum_to_full(eng, orc, curve, group, nshards):
    """msm_amd_multiexp_shard: the partials of all shards add up to the full MSM (multi-GPU decomposition,
    exercised here on one GPU)."""
    import msm_amd
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    n = 1500
    bases, scalars = gu.random_case(curve, group, f"gpu:shard:{curve}:{group}", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    parts = b"".join(C.multiexp_shard(group, bases, scalars, 32, n, s, nshards) for s in range(nshards))
    assert msm_amd.combine(curve, group, parts) == exp


def test_partial_exchange_over_rccl_one_rank(eng, orc):
    """The exchange bench.py uses for N > 1 (msm_amd.dist.PartialExchange: asynchronous all_gather_into_tensor of the 144-byte
    partials on the GPU, one copy back, host addition) on the RCCL backend with a world of one -- what a one-GPU box can run."""
    import socket
    import torch
    import torch.distributed as dist
    import msm_amd
    from msm_amd import dist as mdist
    curve, group, n = "bls12381", 1, 700
    C = eng.curve(curve)
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, "gpu:rccl1", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        part = C.multiexp_shard(group, bases, scalars, 32, n, 0, 1)
        xch = mdist.PartialExchange(curve, group, device=dev)
        h1, h2 = xch.submit(part), xch.submit(part)
        assert xch.result(h1) == exp and xch.result(h2) == exp
        assert mdist.gather_and_combine(curve, group, part, device=dev) == exp
    finally:
        dist.destroy_process_group()
