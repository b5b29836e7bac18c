"""GPU tests added in round 4 (run with -m gpu on the MI355X box):
  * the accumulation's work list (k_items_plan / k_accumulate_items / k_join_items: lanes per bucket in proportion to its population,
    csrc/bucket_kernels.hpp) against the oracle -- every curve and group, forced on at window sizes and lane lengths that exercise one
    lane per bucket, several lanes per bucket, the open-ended top population class, coarser population classes (osh > 0), skewed and
    hot buckets whose rest goes to the tail kernels, sharded and prepared calls;
  * a prepared table released while another thread submits on it, released twice, used after release.
Reference semantics under test: wasmcurves/src/build_multiexp.js:186-233 (every (scalar, point) pair lands in exactly one bucket of
every window) -- the work list only changes which lane adds which entries."""
import json
import os
import random
import subprocess
import sys
import threading

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc(oracle_lib):
    return oracle_lib


def _f(C, group):
    return C.g1m_multiexpAffine if group == 1 else C.g2m_multiexpAffine


@pytest.mark.parametrize("curve,group,n,c,t", [
    ("bls12381", 1, 70000, 0, 0),      # the engine's own choice at a size where the work list is the default path
    ("bls12381", 1, 70000, 14, 8),     # mean 8.5 at T = 8: about half the buckets on two lanes
    ("bls12381", 1, 20000, 9, 16),     # 256 buckets per window, mean 78: every bucket on several lanes, some on all eight (rest -> tail kernels)
    ("bls12381", 1, 50000, 12, 200),   # T = 200: population classes in steps of two (8 T > 1023)
    ("bls12381", 1, 3000, 11, 8),      # most buckets hold 2-3 entries, many are empty
    ("bn128", 1, 120000, 13, 0),
    ("bn128", 1, 9000, 10, 8),
    ("bls12381", 2, 12000, 11, 8),     # a point on a lane pair: 32 items per wave
    ("bn128", 2, 5000, 9, 8),
])
def test_work_list_matches_the_oracle(orc, curve, group, n, c, t):
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("items", 1)
        if c:
            eng.set_window_bits(c)
        if t:
            eng.set_option("items_t", t)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, f"gpu:items:{curve}:{group}:{n}:{c}", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        assert _f(C, group)(bases, scalars, 32, n) == exp
        # two in flight (the second call's sort runs under the first one's work list), and the same call with the work list off
        t1, t2 = C.multiexp_async(group, bases, scalars, 32, n), C.multiexp_async(group, bases, scalars, 32, n)
        assert t1.wait() == exp and t2.wait() == exp
        eng.set_option("items", 2)
        assert _f(C, group)(bases, scalars, 32, n) == exp
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 60000), ("bn128", 1, 30000), ("bls12381", 2, 8000)])
def test_work_list_with_skewed_scalars(orc, curve, group, n):
    """Populations far from uniform under the work list: a witness-like mix (one bucket of window 0 holds most points: it gets its eight
    lanes and the rest is queued as a HOT bucket for the tail kernels), a few dozen distinct scalars (a handful of very long buckets
    per window, all the others empty), every scalar equal, and 64-bit values (every window above the fourth empty)."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("items", 1)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, uniform = gu.random_case(curve, group, f"gpu:items-skew:{curve}:{group}:{n}", n)
        rnd = random.Random(n)
        witness = bytearray(uniform)
        for i in range(n):
            r = rnd.random()
            if r < 0.6:
                witness[32 * i:32 * i + 32] = (1).to_bytes(32, "little")
            elif r < 0.75:
                witness[32 * i:32 * i + 32] = bytes(32)
            elif r < 0.85:
                witness[32 * i + 2:32 * i + 32] = bytes(30)
        few = b"".join(uniform[32 * k:32 * k + 32] for k in (rnd.randrange(40) for _ in range(n)))
        same = uniform[:32] * n
        narrow = b"".join(uniform[32 * i:32 * i + 8] + bytes(24) for i in range(n))
        for name, sc in (("witness", bytes(witness)), ("few", few), ("same", same), ("narrow", narrow)):
            exp = O.normalize(O.multiexp_affine(bases, sc, 32, n))
            for rep in range(2):          # (the second call runs with the skew hint the first one left)
                assert _f(C, group)(bases, sc, 32, n) == exp, (name, rep)
    finally:
        eng.close()


def test_work_list_in_sharded_and_prepared_calls(orc):
    """The windows one rank of a sharded call owns (few buckets, dense: where the sliced launch used to serve) and a prepared table
    (merged windows: one bucket set fed by every window) under the work list."""
    import msm_amd
    curve, group, n = "bls12381", 1, 90000
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("items", 1)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, "gpu:items:sharded", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        for gw in (2, 4, 8):
            parts = b"".join(C.multiexp_shard(group, bases, scalars, 32, n, w, gw) for w in range(gw))
            assert msm_amd.combine(curve, group, parts) == exp, gw
        pb = C.prepare_bases(group, bases, n)
        try:
            assert pb.multiexp(scalars) == exp
        finally:
            pb.release()
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group,n,c,opts", [
    ("bls12381", 1, 300000, 13, {"tile_mode": 2}),                    # 16 coarse bins, spans of 18750 entries: the 1024-lane fine sort reads them once
    ("bls12381", 1, 60000, 18, {"tile_mode": 2, "sort_geom": 2}),     # 512 coarse bins, tiles of 8192 scalars, 256 fine buckets, the 256-lane fine sort
    ("bn128", 1, 60000, 20, {"tile_mode": 2, "sort_geom": 3}),        # 512 bins, 1024 fine buckets, the 1024-lane fine sort
    ("bn128", 1, 60000, 20, {"tile_mode": 2, "sort_geom": 2, "sort_big": 1}),   # 512 bins and the chunk-parallel kernels on 1024 fine buckets
    ("bls12381", 2, 20000, 19, {"tile_mode": 2, "sort_geom": 3}),
    ("bls12381", 1, 60000, 20, {"tile_mode": 2, "sort_geom": 1}),     # the round-3 geometry stays selectable
])
def test_partition_geometries_of_round_4(orc, curve, group, n, c, opts):
    """The bucket sort's new shapes (csrc/engine_call.hpp MsmCall::sort, csrc/sort_kernels.hpp): spans of up to 36864 entries read once by a
    1024-lane workgroup, 512 coarse bins cut from tiles of 8192 scalars (13-bit ranks), the chunk-parallel kernels on 512 bins --
    forced at sizes the oracle can follow (`tile_mode` = 2: the wide tiles these shapes belong to; `sort_geom`), synchronous and two
    calls in flight, on caller-supplied bases and on a prepared table (merged windows: index slices over rows of 8192-padded stride)."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_window_bits(c)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, f"gpu:geom:{curve}:{group}:{n}:{c}", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        assert _f(C, group)(bases, scalars, 32, n) == exp
        t1, t2 = C.multiexp_async(group, bases, scalars, 32, n), C.multiexp_async(group, bases, scalars, 32, n)
        assert t1.wait() == exp and t2.wait() == exp
        pb = C.prepare_bases(group, bases, n)
        try:
            assert pb.info()["table_bytes"] == pb.info()["windows"] * ((n + 8191) // 8192 * 8192) * C.affine_bytes(group)
            assert pb.multiexp(scalars) == exp
        finally:
            pb.release()
        # half the scalars equal: one long span per window, whatever the geometry
        skew = bytearray(scalars)
        for i in range(0, n, 2):
            skew[32 * i:32 * i + 32] = scalars[:32]
        skew = bytes(skew)
        assert _f(C, group)(bases, skew, 32, n) == O.normalize(O.multiexp_affine(bases, skew, 32, n))
    finally:
        eng.close()


def test_release_of_a_prepared_table_is_atomic_with_submissions(orc):
    """msm_amd_release_bases and the submission of a call on the table share one critical section: while one thread
    keeps calls in flight on a table, another thread's release is refused ("still in flight") or, once it succeeds, every later call is
    refused ("released") -- never a call on freed memory.  A second release and a call on the released handle are errors, not crashes."""
    import msm_amd
    curve, group, n = "bn128", 1, 20000
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("slots", 4)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, "gpu:release-race", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        pb = C.prepare_bases(group, bases, n)
        handle = pb.handle
        outcome = {"good": 0, "refused": 0, "bad": []}
        stop = threading.Event()

        def caller():
            while not stop.is_set():
                try:
                    r = pb.multiexp(scalars)
                    if r == exp:
                        outcome["good"] += 1
                    else:
                        outcome["bad"].append("wrong result")
                except msm_amd.MsmAmdError as e:
                    if "released" in str(e) or "not bases of this context" in str(e):
                        outcome["refused"] += 1
                        return
                    outcome["bad"].append(str(e))
                    return

        th = threading.Thread(target=caller)
        th.start()
        import time
        released, refusals, t0 = False, 0, time.time()
        while outcome["good"] < 5 and time.time() - t0 < 60:      # let a few calls through first
            time.sleep(0.001)
        while time.time() - t0 < 120:
            rc = eng.L.msm_amd_release_bases(eng.ctx, handle)
            if rc == 0:
                released = True
                break
            refusals += 1
            assert "in flight" in eng.L.msm_amd_last_error(eng.ctx).decode()
        stop.set()
        th.join(timeout=120)
        assert not th.is_alive() and not outcome["bad"], outcome
        assert released and outcome["good"] > 0
        pb.handle = None
        # the handle is gone: a second release and a call on it are refused without touching it
        assert eng.L.msm_amd_release_bases(eng.ctx, handle) == msm_amd.EINVAL
        import ctypes
        buf = ctypes.create_string_buffer(96)
        assert eng.L.msm_amd_multiexp_prepared(eng.ctx, handle, 0, scalars, 32, n, buf) == msm_amd.EINVAL
        assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp        # the context is fine
    finally:
        eng.close()


def test_a_nearly_full_device_gives_enomem_or_the_right_bytes():
    """In a freshly started child process (tests/oom_worker.py): a context is created, the device is filled to ~200 MB free and then to
    ~24 MB free, and G2 and G1 calls with queued segments -- the tail-combine kernels doing real work for the first time in the process --
    are issued at both levels, plus the entry points whose kernels need more scratch than the context reserved.  Before round 4 the
    HSA runtime ABORTED the process there (a queue's scratch could not be allocated at the dispatch).  Now the MSM path's kernels carry
    no private segment beyond spills, their scratch is reserved by msm_amd_create, and the other entry points ask before they launch:
    every call returns the right bytes or MSM_AMD_ENOMEM, and the child exits 0."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "oom_worker.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, f"the child died ({p.returncode})\nstdout:\n{p.stdout[-3000:]}\nstderr:\n{p.stderr[-3000:]}"
    res = json.loads(p.stdout.strip().splitlines()[-1])
    log = os.path.join(root, "gpurun_out")
    if os.path.isdir(log):
        with open(os.path.join(log, "r04_oom_child.txt"), "w") as f:
            f.write("\n".join(res["notes"]) + "\n")
    assert res["ok"], res


def _narrow(raw, n, bits):
    """n 32-byte scalars of `bits` significant bits (the top one set: the width is exact) from the bytes of `raw`"""
    out = bytearray(raw[:32 * n])
    for i in range(n):
        nb = bits // 8
        out[32 * i + nb + (1 if bits % 8 else 0):32 * i + 32] = bytes(32 - nb - (1 if bits % 8 else 0))
        if bits % 8:
            out[32 * i + nb] &= (1 << (bits % 8)) - 1
        out[32 * i + (bits - 1) // 8] |= 1 << ((bits - 1) % 8)
    return bytes(out)


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 30000), ("bn128", 1, 120000), ("bls12381", 2, 4000)])
def test_windows_follow_the_observed_scalar_width(orc, curve, group, n):
    """Scalars much narrower than their 32 bytes (128-bit challenges, 64-bit witness values, a width the window does not divide, a 253-bit
    field): the histogram pass reports the widest scalar of every call, the maximum over the last eight calls is the width the next calls'
    windows tile (fewer windows, no carry bucket: msm_amd_get_geometry shows it), and whatever comes next is still exact -- a few bits
    more (overflow window), a much wider call (reported by the histogram pass, run again over the declared width when collected: the
    caller only sees the right bytes; from then on the hint is the wider width), alternating widths (the wide layout stays, no second
    runs), a stream that turns narrower (follows eight calls later), asynchronous calls in flight, option "narrow" = 0."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, uniform = gu.random_case(curve, group, f"gpu:narrow:{curve}:{group}:{n}", n)
        f = _f(C, group)
        memo = {}

        def check(sc, what):
            if sc not in memo:
                memo[sc] = O.normalize(O.multiexp_affine(bases, sc, 32, n))
            assert f(bases, sc, 32, n) == memo[sc], what
            return eng.geometry()["windows"], eng.hints()["narrow_bits"]

        full_w, _ = check(uniform, "full width, fresh context")
        for bits in (128, 64, 200, 13, 253):
            eng.set_option("narrow", 1)                          # (forgets what was observed: every width starts from nothing)
            sc = _narrow(uniform, n, bits)
            w1, h1 = check(sc, f"{bits} bits, first call")
            w2, h2 = check(sc, f"{bits} bits, second call")
            w3, h3 = check(sc, f"{bits} bits, third call")
            assert h1 == 0 and h2 == bits and h3 == bits and w3 <= full_w and (w3 < full_w or bits > 240), (bits, w1, w2, w3, h1, h2, h3)
            # three bits more than the hint: the overflow window takes them, exactly, and the hint follows
            wider = min(bits + 3, 256)
            w4, h4 = check(_narrow(uniform, n, wider), f"{bits} bits + a few")
            assert h4 == wider or (wider >= 254 and h4 >= 254), (bits, h4)
        # a stream that turns much wider: ONE call is run again inside its wait, then the hint is the wider width; turning narrower again takes eight calls
        eng.set_option("narrow", 1)
        s64, s128 = _narrow(uniform, n, 64), _narrow(uniform, n, 128)
        for k in range(3):
            w64, h = check(s64, f"64-bit stream {k}")
        assert h == 64
        w, h = check(s128, "the first 128-bit call on a context that expects 64 bits")
        assert h == 128 and w > w64
        w128, h = check(s128, "the second 128-bit call")
        assert h == 128 and w128 < full_w
        for k in range(7):
            w, h = check(s64, f"64-bit calls again {k}")
            assert h == 128 and w == w128, (k, w, h)
        w, h = check(s64, "the eighth: the 128-bit calls have left the last eight")
        assert h == 64
        assert check(s64, "narrow layout again")[0] == w64
        # a full-width call on a context that expects narrow ones, and alternating widths afterwards: the wide layout stays
        w, h = check(uniform, "full width after 64-bit calls")
        assert h >= 254 or h == 0
        for k in range(6):
            w, h = check(s64 if k % 2 else uniform, f"alternating {k}")
            assert w >= full_w - 1, (k, w, h)
        # asynchronous: a narrow and a wide call in flight, collected in the other order
        eng.set_option("narrow", 1)
        check(s64, "narrow")
        check(s64, "narrow")
        t1, t2 = C.multiexp_async(group, bases, s64, 32, n), C.multiexp_async(group, bases, uniform, 32, n)
        assert t2.wait() == memo[uniform] and t1.wait() == memo[s64]
        eng.set_option("narrow", 0)
        for _ in range(3):
            w, h = check(s64, "adaptation off")
            assert h == 0 and w == full_w
    finally:
        eng.close()


def _seeds(default):
    """the committed seed, plus any in MSM_TEST_EXTRA_SEEDS (comma-separated; soak runs)"""
    extra = [int(x) for x in os.environ.get("MSM_TEST_EXTRA_SEEDS", "").split(",") if x.strip()]
    return [default] + extra


@pytest.mark.parametrize("seed", _seeds(20261005))
def test_randomised_round4_configurations(orc, seed):
    """50 seeded random combinations of what round 4 added or touched: the work list on / off / forced with random lane lengths, every
    partition geometry with either tile size, wide and narrow windows, scalar sizes the window does or does not divide, uniform / full-width /
    short / heavily repeated scalars (long and hot buckets: the inlined tail kernels), both forms of the bucket reduction with random
    running-sum lengths, synchronous, two in flight, prepared tables and hybrid shards summed on this GPU -- each against the oracle."""
    import msm_amd
    from msm_amd import dist as mdist
    rnd = random.Random(seed)
    eng = msm_amd.Engine(0)
    try:
        for it in range(50):
            curve = rnd.choice(["bls12381", "bn128"])
            group = 1 if rnd.random() < 0.75 else 2
            n = rnd.choice([1, 5, 64, 300, 2049, 6000, 20000, 45000] if group == 1 else [1, 9, 130, 700, 3000])
            ss = rnd.choice([32, 32, 32, 31, 16, 8, 33])
            C = eng.curve(curve)
            O = orc.Oracle(curve, group)
            asz = C.affine_bytes(group)
            bases, _ = gu.random_case(curve, group, f"gpu:rand4:{seed}:{it}", n)
            raw = gu.stream(f"gpu:rand4:{seed}:{it}:s", n * ss)
            kind = rnd.choice(["uniform", "uniform", "full", "short", "repeated"])
            sc = bytearray(raw)
            if kind == "full":
                for i in range(n):
                    sc[i * ss + ss - 1] |= 0x80
            elif kind == "short":
                for i in range(n):
                    sc[i * ss + max(1, ss // 2):(i + 1) * ss] = bytes(ss - max(1, ss // 2))
            elif kind == "repeated":
                k = rnd.choice([1, 3, 17])
                for i in range(n):
                    if rnd.random() < 0.8:
                        j = rnd.randrange(k)
                        sc[i * ss:(i + 1) * ss] = raw[j * ss:(j + 1) * ss] if j < n else raw[:ss]
            scalars = bytes(sc)
            opts = {"items": rnd.choice([0, 1, 1, 2]), "items_t": rnd.choice([0, 0, 8, 9, 16, 40, 300]), "sort_geom": rnd.choice([0, 0, 1, 2, 3]),
                    "tile_mode": rnd.choice([0, 1, 2, 2]), "sort_big": rnd.choice([0, 0, 1, 2]), "hot": rnd.choice([1, 1, 0]),
                    "reduce_planes": rnd.choice([1, 1, 2, 2, 0]), "reduce_l": rnd.choice([0, 0, 0, 2, 4, 8, 32])}
            for k, v in opts.items():
                eng.set_option(k, v)
            c = rnd.choice([0, 0, 4, 9, 11, 13, 14, 16, 18, 19, 20, 21])
            eng.set_window_bits(c)
            exp = gu.canon(curve, group, O.normalize(O.multiexp_affine(bases, scalars, ss, n)), O.zero())
            mode = rnd.choice(["full", "full", "async2", "prepared", "hybrid"])
            tag = (seed, it, curve, group, n, ss, kind, c, mode, opts)
            if mode == "full":
                got = C._multiexp(group, 0, bases, scalars, ss, n)
            elif mode == "async2":
                t1, t2 = C.multiexp_async(group, bases, scalars, ss, n), C.multiexp_async(group, bases, scalars, ss, n)
                got = t1.wait()
                assert t2.wait() == got, tag
            elif mode == "prepared":
                pb = C.prepare_bases(group, bases, n, scalar_bits=rnd.choice([8 * ss, min(8 * ss + 5, 512)]))
                try:
                    got = pb.multiexp(scalars, ss, n)
                finally:
                    pb.release()
            else:
                world = rnd.choice([2, 3, 4, 8])
                gw = rnd.choice([w for w in range(1, world + 1) if world % w == 0])
                parts = []
                for r in range(world):
                    first, count = mdist.slice_of(n, world // gw, r // gw)
                    parts.append(C.multiexp_shard(group, bases[first * asz:(first + count) * asz], scalars[first * ss:(first + count) * ss], ss, count, r % gw, gw))
                got = msm_amd.combine(curve, group, b"".join(parts))
            assert got == exp, (tag, eng.geometry(), eng.hints())
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group,n,c,rl", [
    ("bls12381", 1, 4000, 9, 4),       # 256 buckets per window: one wave per window (no second stage)
    ("bls12381", 1, 4000, 10, 4),      # two waves per window: the smallest second stage
    ("bls12381", 1, 6000, 16, 2),      # 256 waves per window: every lane of the second stage adds four waves first, two upper planes
    ("bls12381", 1, 6000, 16, 8),      # 64 waves per window: the second stage's wave exactly full
    ("bls12381", 1, 6000, 16, 128),    # long running sums, four waves per window
    ("bls12381", 1, 3000, 19, 32),     # 2^18 buckets per window
    ("bn128", 1, 5000, 13, 4),
    ("bn128", 1, 5000, 16, 2),
    ("bls12381", 2, 2500, 12, 4),      # a point on a lane pair: 32 points per wave
    ("bls12381", 2, 2500, 16, 16),     # ... 64 waves per window on a 32-lane second stage: two per lane, one upper plane
    ("bn128", 2, 2500, 14, 8),
])
def test_reduction_as_plain_sums(orc, curve, group, n, c, rl):
    """The bucket reduction without a scalar multiplication (k_bucket_planes / k_planes_combine, csrc/bucket_kernels.hpp: the window sum as
    sum T_t + sum_b 2^(b + log2 L) P_b, the weights applied by the host's Horner pass) against the oracle, forced on at window sizes and
    running-sum lengths that give one, two, a full wave of and four times a full wave of waves per window -- and the same calls through
    k_bucket_reduce.  Reference semantics: wasmcurves/src/build_multiexp.js:186-233 (the window's bucket sum)."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_option("reduce_planes", 2)
        eng.set_option("reduce_l", rl)
        eng.set_window_bits(c)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, f"gpu:planes:{curve}:{group}:{n}:{c}", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        assert _f(C, group)(bases, scalars, 32, n) == exp
        t1, t2 = C.multiexp_async(group, bases, scalars, 32, n), C.multiexp_async(group, bases, scalars, 32, n)
        assert t1.wait() == exp and t2.wait() == exp
        # every scalar equal: one bucket per window holds everything, all other sums of the butterfly are infinity
        same = scalars[:32] * n
        exp_same = O.normalize(O.multiexp_affine(bases, same, 32, n))
        assert _f(C, group)(bases, same, 32, n) == exp_same
        # 64-bit values: the windows above them are empty (their partials are written as infinities by whichever stage runs last)
        small = b"".join(scalars[i * 32:i * 32 + 8] + bytes(24) for i in range(n))
        exp_small = O.normalize(O.multiexp_affine(bases, small, 32, n))
        eng.set_option("narrow", 0)
        assert _f(C, group)(bases, small, 32, n) == exp_small
        eng.set_option("reduce_planes", 0)
        assert _f(C, group)(bases, scalars, 32, n) == exp
    finally:
        eng.close()


def test_reduction_follows_the_reported_density(orc):
    """reduce(): a pipelined call of the headline's shape (16-bit windows, >= 2^19 buckets) keeps k_bucket_reduce only while its buckets are
    expected to be full; once a call of the same shape has reported fewer than 24 entries per bucket, the next ones in flight take the plane
    kernels.  Both give the oracle's bytes, in either order, with one or two calls in flight."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_window_bits(16)
        C = eng.curve("bls12381")
        O = orc.Oracle("bls12381", 1)
        n = 6000
        bases, scalars = gu.random_case("bls12381", 1, "gpu:density", n)
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        ts = [C.multiexp_async(1, bases, scalars, 32, n) for _ in range(2)]     # nothing reported yet: the second one is "hidden" -> k_bucket_reduce
        assert [t.wait() for t in ts] == [exp, exp]
        for _ in range(3):                                                       # 0.2 entries per bucket reported: plane kernels for calls in flight
            ts = [C.multiexp_async(1, bases, scalars, 32, n) for _ in range(2)]
            assert [t.wait() for t in ts] == [exp, exp]
        eng.set_option("reduce_planes", 0)
        ts = [C.multiexp_async(1, bases, scalars, 32, n) for _ in range(2)]
        assert [t.wait() for t in ts] == [exp, exp]
    finally:
        eng.close()


@pytest.mark.parametrize("curve,group,n", [("bls12381", 1, 6000), ("bn128", 1, 9000), ("bls12381", 2, 2500), ("bn128", 2, 3000)])
def test_deferred_reduction_in_bursts_and_alone(orc, curve, group, n):
    """Option defer_reduce (csrc/engine_call.hpp, submit_msm): a call in a stream of calls leaves its bucket reduction
    (wasmcurves/src/build_multiexp.js:120-184, reduceTable) to the next submission or to the wait for its own ticket, which also picks the
    reduction's form.  16-bit windows put the accumulation on the context's shared stream at any n, the regime the deferral applies to.
    (G2 calls are never deferred: their cases run the same sequences on the undeferred path.)
    Every order of submissions and waits must give the oracle's bytes, with the option on and off."""
    import msm_amd
    eng = msm_amd.Engine(0)
    try:
        eng.set_window_bits(16)
        eng.set_option("slots", 3)
        C = eng.curve(curve)
        O = orc.Oracle(curve, group)
        asz = len(gu.random_case(curve, group, "probe", 1)[0])
        bases, scalars = gu.random_case(curve, group, f"gpu:defer:{curve}:{group}:{n}", n)
        m = n // 3
        exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
        exp_m = O.normalize(O.multiexp_affine(bases[:m * asz], scalars[:32 * m], 32, m))
        sync = _f(C, group)
        for defer in (1, 0, 1):
            eng.set_option("defer_reduce", defer)
            assert sync(bases, scalars, 32, n) == exp                      # a lone call before any burst
            # a burst of three, waited for newest first: the newest one's reduction is queued by its own wait with two older calls in flight
            t1, t2, t3 = C.multiexp_async(group, bases, scalars, 32, n), C.multiexp_async(group, bases[:m * asz], scalars[:32 * m], 32, m), C.multiexp_async(group, bases, scalars, 32, n)
            assert t3.wait() == exp and t1.wait() == exp and t2.wait() == exp_m
            # the first call after a burst is deferred although nothing is in flight (the context has seen a burst lately): alone, then followed
            assert sync(bases[:m * asz], scalars[:32 * m], 32, m) == exp_m
            t1 = C.multiexp_async(group, bases, scalars, 32, n)
            t2 = C.multiexp_async(group, bases[:m * asz], scalars[:32 * m], 32, m)    # queues t1's reduction in the hidden form
            assert t1.wait() == exp
            t3 = C.multiexp_async(group, bases, scalars, 32, n)
            assert t2.wait() == exp_m and t3.wait() == exp
            # a steady stream, two in flight, and then enough lone calls to run the context's memory of the burst down
            fl, got = [], []
            for i in range(7):
                fl.append(C.multiexp_async(group, bases, scalars, 32, n))
                if len(fl) == 2:
                    got.append(fl.pop(0).wait())
            got += [t.wait() for t in fl]
            assert all(g == exp for g in got)
            for i in range(10):
                assert sync(bases[:m * asz], scalars[:32 * m], 32, m) == exp_m
    finally:
        eng.close()


@pytest.mark.parametrize("seed", [11, 12, 13, 14] + [int(s) for s in os.environ.get("MSM_TEST_EXTRA_SEEDS", "").split(",") if s.strip()])
def test_deferred_reduction_under_random_orders_of_submissions_and_waits(orc, seed):
    """Seeded sequences of asynchronous submissions (three sizes, G1 of both curves on one context, up to three in flight), waits for a
    randomly chosen ticket, synchronous calls in between and changes of option defer_reduce: whichever call's reduction is queued by whichever
    event (the next submission, its own wait with older or no calls in flight), every result is the oracle's."""
    import msm_amd
    rnd = random.Random(seed)
    eng = msm_amd.Engine(0)
    try:
        eng.set_window_bits(16)
        eng.set_option("slots", 3)
        cases = []
        for curve in ("bls12381", "bn128"):
            C, O = eng.curve(curve), orc.Oracle(curve, 1)
            asz = len(gu.random_case(curve, 1, "probe", 1)[0])
            for n in (700, 2500, 5000):
                b, s = gu.random_case(curve, 1, f"gpu:defer-rnd:{curve}:{n}:{seed}", n)
                cases.append((C, b, s, n, O.normalize(O.multiexp_affine(b, s, 32, n))))
        flying = []
        for step in range(60):
            r = rnd.random()
            if r < 0.05:
                eng.set_option("defer_reduce", rnd.choice((0, 1, 2)))
            elif r < 0.15 and len(flying) < 3:
                C, b, s, n, exp = rnd.choice(cases)
                assert C.g1m_multiexpAffine(b, s, 32, n) == exp
            elif r < 0.65 and len(flying) < 3:
                C, b, s, n, exp = rnd.choice(cases)
                flying.append((C.multiexp_async(1, b, s, 32, n), exp))
            elif flying:
                t, exp = flying.pop(rnd.randrange(len(flying)))
                assert t.wait() == exp
        while flying:
            t, exp = flying.pop(rnd.randrange(len(flying)))
            assert t.wait() == exp
    finally:
        eng.close()
