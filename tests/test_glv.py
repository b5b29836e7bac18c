"""GLV pre-pass for BLS12-381 G1 (reference wasmcurves/src/build_glv.js)."""
import json
import os

import pytest

import golden_util as gu


def _kat():
    with open(os.path.join(gu.GOLDEN, "kats_glv.json")) as f:
        return json.load(f)


def _edge_scalars():
    r = gu.R_ORDER["bls12381"]
    vals = [0, 1, 2, r - 1, r, r + 1, 2 * r, 2 * r + 5, (1 << 256) - 1, 1 << 255, (1 << 128) - 1, 1 << 128, int(_kat()["scalar"])]
    raw = gu.stream("glv:scalars", 32 * 200)
    vals += [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(200)]
    vals += [int.from_bytes(raw[32 * i:32 * i + 32], "little") % r for i in range(200)]
    return vals


def test_oracle_glv_matches_reference_kat():
    from oracle import glv
    k = _kat()
    k1, k2, sign = glv.decompose_scalar(int(k["scalar"]))
    assert [str(k1), str(k2)] == k["expected"] and sign == k["sign"]
    # the identity the reference's "scalarMul is correct" test checks (test/glv.js:67-101): k = +-k1 +- k2*lambda (mod r)
    for kk in _edge_scalars():
        a, b, s = glv.decompose_scalar(kk)
        s1 = a if s & 1 else -a
        s2 = b if s & 2 else -b
        if a < (1 << 127) and b < (1 << 127):
            assert (s1 + s2 * glv.LAMBDA - kk) % glv.R == 0


def test_host_decompose_and_endomorphism_match_oracle(oracle_lib):
    import msm_amd
    from oracle import glv
    O = oracle_lib.Oracle("bls12381", 1)
    for kk in _edge_scalars():
        assert msm_amd.glv_decompose_scalar(kk.to_bytes(32, "little")) == glv.decompose_scalar(kk)
    pts = gu.pool_points("bls12381", 1, [0, 1, 2, 3])
    for i in range(4):
        p = pts[96 * i:96 * i + 96]
        for sign in (0, 1):
            assert msm_amd.glv_endomorphism(p, sign) == glv.endomorphism(O, p, sign)
    # phi(P) = lambda * P
    gen = O.to_jacobian(pts[:96])
    lamP = O.to_affine(O.times_scalar(gen, glv.LAMBDA.to_bytes(32, "little")))
    assert msm_amd.glv_endomorphism(pts[:96], 1) == lamP


def test_engine_split_identity_and_bounds():
    """The split the engine applies inside BLS12-381 G1 MSMs: k == +-|k1| +- k2*lambda (mod r), both halves below 2^127
    (so that their signed radix-2^16 digits never carry out of the eighth window)."""
    import msm_amd
    from oracle import glv
    r = glv.R
    extra = [(r - 1) // 2, (r + 1) // 2, glv.LAMBDA - 1, glv.LAMBDA, glv.LAMBDA + 1, glv.LAMBDA // 2, glv.LAMBDA // 2 + 1, r - glv.LAMBDA]
    for kk in _edge_scalars() + extra:
        a, b, s = msm_amd.glv_split(kk.to_bytes(32, "little"))
        assert a < (1 << 127) and b < (1 << 127), hex(kk)
        s1 = -a if s & 1 else a
        s2 = -b if s & 2 else b
        assert (s1 + s2 * glv.LAMBDA - kk) % r == 0, hex(kk)


@pytest.mark.gpu
def test_gpu_preprocess_matches_oracle_and_preserves_the_msm(oracle_lib):
    """g1m_glv_preprocessEndomorphism on the GPU == the oracle's, byte for byte; and the MSM of the 2n preprocessed
    pairs equals the MSM of the n original pairs (the reference's test/glv.js:103-191)."""
    import msm_amd
    from oracle import glv
    O = oracle_lib.Oracle("bls12381", 1)
    eng = msm_amd.Engine(0)
    try:
        C = eng.curve("bls12381")
        vals = _edge_scalars()
        n = len(vals)
        bases, _ = gu.random_case("bls12381", 1, "gpu:glv", n)
        bases = bytes(96) + bases[96:]                       # an affine infinity among the inputs
        scalars = b"".join(v.to_bytes(32, "little") for v in vals)
        po, so = C.g1m_glv_preprocessEndomorphism(bases, scalars, n)
        epo, eso = glv.preprocess(O, bases, scalars, n)
        assert so == eso
        assert po == epo
        # scalars reduced mod r so that the 128-bit magnitudes are exact: same group element
        red = b"".join((v % glv.R).to_bytes(32, "little") for v in vals)
        po, so = C.g1m_glv_preprocessEndomorphism(bases, red, n)
        assert C.g1m_multiexpAffine(po, so, 32, 2 * n) == C.g1m_multiexpAffine(bases, red, 32, n)
        assert C.g1m_multiexpAffine(po, so, 32, 2 * n) == O.normalize(O.multiexp_affine(bases, red, 32, n))
        with pytest.raises(msm_amd.MsmAmdError):
            eng.curve("bn128").g1m_glv_preprocessEndomorphism(bases, scalars, n)
    finally:
        eng.close()
