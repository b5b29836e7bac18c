"""CPU restatement of the reference's GLV pre-pass for BLS12-381 G1 (TEST INFRASTRUCTURE ONLY).

Follows wasmcurves/src/build_glv.js with Python integers:
  constants            :15-22
  _decomposeScalar     :53-146   (q1 = k // r, q2 = k*negV1 // r, k1 = k - q1 - q2*u0, k2 = q1*negV1 - q2;
                                  magnitudes truncated to 128 bits as the reference stores two i64 words each)
  _endomorphism        :150-163
  _preprocessEndomorphism :178-267
Pinned by the reference's own known answer (wasmcurves/test/glv.js:50-65, tests/golden/kats_glv.json); the
functions are not exported by the prebuilt .wasm, so there is no run-time vector for them.
"""
R = 52435875175126190479447740508185965837690552500527637822603658699938581184513
NEG_V1 = 228988810152649578064853576960394133503
U0 = 228988810152649578064853576960394133504
BETA = 793479390729215512621379701633421447060886740281060493010456487427281649075476305620758731620350
Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
LAMBDA = (-U0) % R      # phi(P) = (beta*x, y) = LAMBDA * P on the r-torsion


def decompose_scalar(k):
    q1 = k // R
    q2 = (k * NEG_V1) // R
    k1 = k - q1 - q2 * U0
    k2 = q1 * NEG_V1 - q2
    sign = (1 if k1 >= 0 else 0) | (2 if k2 >= 0 else 0)
    mask = (1 << 128) - 1
    return abs(k1) & mask, abs(k2) & mask, sign


def endomorphism(oracle_g1, point_affine, sign):
    """oracle_g1: oracle.Oracle('bls12381', 1) (field ops); point_affine: 96 bytes"""
    beta_m = oracle_g1.to_montgomery(BETA.to_bytes(48, "little"))
    x, y = point_affine[:48], point_affine[48:]
    bx = oracle_g1.f_mul(x, beta_m)
    return bx + (y if sign else oracle_g1.f_neg(y))


def preprocess(oracle_g1, points, scalars, n):
    po, so = bytearray(), bytearray()
    for i in range(n):
        p = points[96 * i:96 * i + 96]
        k = int.from_bytes(scalars[32 * i:32 * i + 32], "little")
        k1, k2, sign = decompose_scalar(k)
        x, y = p[:48], p[48:]
        po += x + (y if sign & 1 else oracle_g1.f_neg(y))
        po += endomorphism(oracle_g1, p, sign >> 1)
        so += k1.to_bytes(32, "little") + k2.to_bytes(32, "little")
    return bytes(po), bytes(so)
