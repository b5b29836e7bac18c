"""Helpers shared by the parity tests: the SHA-256 counter stream that oracle/gen_golden.js uses
to derive inputs, the committed point pools, and curve constants."""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

R_ORDER = {
    "bls12381": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
    "bn128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
}
Q_MOD = {
    "bls12381": 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787,
    "bn128": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}
N8 = {"bls12381": 48, "bn128": 32}


def stream(label, nbytes):
    out = bytearray()
    k = 0
    while len(out) < nbytes:
        out += hashlib.sha256(f"{label}:{k}".encode()).digest()
        k += 1
    return bytes(out[:nbytes])


def scalars_for(curve, label, n, scalar_size=32):
    raw = stream(label + ":s", 32 * n)
    r = R_ORDER[curve]
    out = bytearray()
    for i in range(n):
        v = int.from_bytes(raw[32 * i:32 * i + 32], "little") % r
        b = v.to_bytes(32, "little")
        out += (b + bytes(max(0, scalar_size - 32)))[:scalar_size]
    return bytes(out)


def indices_for(label, n, pool_count):
    raw = stream(label + ":p", 4 * n)
    return [int.from_bytes(raw[4 * i:4 * i + 4], "little") % pool_count for i in range(n)]


_cache = {}


def golden(curve):
    if curve not in _cache:
        with open(os.path.join(GOLDEN, f"golden_{curve}.json")) as f:
            _cache[curve] = json.load(f)
    return _cache[curve]


def pool(curve, group):
    key = (curve, group, "pool")
    if key not in _cache:
        G = golden(curve)["groups"][f"g{group}"]
        with open(os.path.join(GOLDEN, G["pool_file"]), "rb") as f:
            _cache[key] = f.read()
    return _cache[key]


def pool_points(curve, group, idx):
    asz = 2 * N8[curve] * group
    p = pool(curve, group)
    return b"".join(p[i * asz:(i + 1) * asz] for i in idx)


def msm_inputs(curve, group, rec):
    """(bases, scalars) bytes of a golden MSM record (explicit, or re-derived from its label)."""
    if "bases" in rec:
        return bytes.fromhex(rec["bases"]), bytes.fromhex(rec["scalars"])
    G = golden(curve)["groups"][f"g{group}"]
    idx = indices_for(rec["label"], rec["n"], G["pool_count"])
    return pool_points(curve, group, idx), scalars_for(curve, rec["label"], rec["n"], rec["scalar_size"])


def random_case(curve, group, label, n, scalar_size=32):
    """Seeded (bases, scalars) drawn from the committed pool; used by oracle-vs-GPU tests."""
    G = golden(curve)["groups"][f"g{group}"]
    idx = indices_for(label, n, G["pool_count"])
    return pool_points(curve, group, idx), scalars_for(curve, label, n, scalar_size)


def canon(curve, group, jac_norm, zero):
    """Canonical form of a *normalised* reference output.  The reference has two byte images for the
    identity: (0, R, 0) from g?m_zero and (0, 0, R) from addMixed(infinity, affine infinity)
    (build_curve_jacobian_a0.js:697-704), which all its own operations (incl. g?m_eq) treat alike.
    The engine always writes (0, R, 0)."""
    fsz = N8[curve] * group
    if jac_norm[:2 * fsz] == bytes(2 * fsz) or jac_norm[2 * fsz:] == bytes(fsz):
        return zero
    return jac_norm
