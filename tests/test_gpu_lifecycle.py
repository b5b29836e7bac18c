"""Life-cycle cases of a context on the GPU: a group left and joined again, two contexts of one GPU driven from two threads at once,
a context closed with calls still in flight.  Each result against the oracle."""
import os
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "zprize-wasm-msm_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    return o


def test_a_group_can_be_left_and_joined_again(orc):
    """msm_amd_dist_init / dist_destroy twice on one context (a group of one: RCCL all-gather of a single partial), calls of the group
    and plain calls in between, a ticket of the first group unusable in the second."""
    import msm_amd
    curve, group, n = "bls12381", 1, 3000
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, "gpu:lifecycle:dist", n)
    exp = O.normalize(O.multiexp_affine(bases, scalars, 32, n))
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve(curve)
        for round_ in range(2):
            try:
                eng.dist_init(0, 1, msm_amd.dist_unique_id())
            except msm_amd.MsmAmdError as e:
                pytest.skip(f"RCCL not usable here: {e}")
            assert eng.dist_info() == (0, 1)
            assert C.multiexp_dist(group, bases, scalars, 32, n) == exp
            t = C.multiexp_dist_async(group, bases, scalars, 32, n)
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp        # a plain call while a group call is in flight
            assert t.wait() == exp
            eng.dist_destroy()
            assert eng.dist_info() == (0, 0)
            with pytest.raises(msm_amd.MsmAmdError):
                C.multiexp_dist(group, bases, scalars, 32, n)                 # not in a group any more
            assert C.g1m_multiexpAffine(bases, scalars, 32, n) == exp
    finally:
        eng.close()


def test_two_contexts_of_one_gpu_from_two_threads(orc):
    """Two engines on the same device, each driven by its own thread with pipelined calls of different curves: the streams and scratch
    of one never touch the other's."""
    import msm_amd
    jobs = []
    for k, (curve, group, n) in enumerate([("bls12381", 1, 20000), ("bn128", 1, 30000)]):
        O = orc.Oracle(curve, group)
        bases, scalars = gu.random_case(curve, group, f"gpu:lifecycle:two:{k}", n)
        m = 1500                                                              # the oracle on a prefix, the engines on prefix and whole
        jobs.append((curve, group, n, bases, scalars, m, O.normalize(O.multiexp_affine(bases[:m * 2 * O.fsz], scalars[:32 * m], 32, m))))
    errors, wholes = [], [None, None]

    def worker(k):
        curve, group, n, bases, scalars, m, exp_prefix = jobs[k]
        try:
            eng = msm_amd.Engine(0)
            try:
                C = eng.curve(curve)
                asz = C.affine_bytes(group)
                ref = C._multiexp(group, 0, bases, scalars, 32, n)
                for rep in range(6):
                    t1 = C.multiexp_async(group, bases, scalars, 32, n)
                    t2 = C.multiexp_async(group, bases[:m * asz], scalars[:32 * m], 32, m)
                    if t1.wait() != ref or t2.wait() != exp_prefix:
                        errors.append((k, rep))
                wholes[k] = ref
            finally:
                eng.close()
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    # the whole results once more on a fresh context, one at a time
    eng = msm_amd.Engine(0)
    try:
        for k, (curve, group, n, bases, scalars, _m, _e) in enumerate(jobs):
            assert eng.curve(curve)._multiexp(group, 0, bases, scalars, 32, n) == wholes[k]
    finally:
        eng.close()


def test_closing_a_context_with_calls_in_flight(orc):
    """msm_amd_destroy waits for the GPU work of tickets nobody collected (their inputs must stay alive until then -- they do here) and
    frees everything; the device is fine afterwards."""
    import msm_amd
    curve, group, n = "bls12381", 1, 50000
    O = orc.Oracle(curve, group)
    bases, scalars = gu.random_case(curve, group, "gpu:lifecycle:close", n)
    eng = msm_amd.Engine(0)
    C = eng.curve(curve)
    pb = C.prepare_bases(group, bases, n)
    tickets = [C.multiexp_async(group, bases, scalars, 32, n), pb.multiexp_async(scalars)]
    eng.close()                                                               # two calls in flight, one on an unreleased table
    del tickets
    eng2 = msm_amd.Engine(0)
    try:
        m = 2000
        assert eng2.curve(curve).g1m_multiexpAffine(bases[:96 * m], scalars[:32 * m], 32, m) == O.normalize(O.multiexp_affine(bases[:96 * m], scalars[:32 * m], 32, m))
    finally:
        eng2.close()
