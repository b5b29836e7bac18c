"""ctypes loader for the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (zprize-wasm-msm_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB: another build of the same source (the sanitizer leg of the CPU tests loads liboracle_san.so through it)
_LIB = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

CURVE_ID = {"bls12381": 0, "bn128": 1}
N8 = {"bls12381": 48, "bn128": 32}


def build(force=False):
    """Compile oracle.c -> liboracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_LIB)], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _buf(n):
    return ctypes.create_string_buffer(n)


class Oracle:
    """Reference-shaped view of one curve/group: methods named after the WASM exports."""

    def __init__(self, curve, group):
        self.curve, self.group = curve, group
        self.cid = CURVE_ID[curve]
        self.n8 = N8[curve]
        self.fsz = self.n8 * group           # field element bytes
        self.asz = 2 * self.fsz               # affine point bytes
        self.jsz = 3 * self.fsz               # Jacobian point bytes
        self.L = lib()

    # --- MSM (build_multiexp.js:251-371, :96-249) ---
    def multiexp(self, bases, scalars, scalar_size, n, affine=True):
        out = _buf(self.jsz)
        rc = self.L.oracle_multiexp(self.cid, self.group, int(affine), bytes(bases), bytes(scalars), scalar_size, n, out)
        assert rc == 0, rc
        return out.raw

    def multiexp_affine(self, bases, scalars, scalar_size, n):
        return self.multiexp(bases, scalars, scalar_size, n, True)

    def multiexp_chunk(self, bases, scalars, scalar_size, n, start_bit, chunk_size, affine=True):
        out = _buf(self.jsz)
        rc = self.L.oracle_multiexp_chunk(self.cid, self.group, int(affine), bytes(bases), bytes(scalars), scalar_size, n,
                                          start_bit, chunk_size, out)
        assert rc == 0, rc
        return out.raw

    def get_chunk(self, scalar, start_bit, chunk_size):
        self.L.oracle_get_chunk.restype = ctypes.c_uint32
        return self.L.oracle_get_chunk(bytes(scalar), len(scalar), start_bit, chunk_size)

    # --- group ops ---
    def _g2(self, fn, a, b, outsz):
        out = _buf(outsz)
        assert getattr(self.L, fn)(self.cid, self.group, bytes(a), bytes(b), out) == 0
        return out.raw

    def _g1(self, fn, a, outsz):
        out = _buf(outsz)
        assert getattr(self.L, fn)(self.cid, self.group, bytes(a), out) == 0
        return out.raw

    def add(self, a, b): return self._g2("oracle_g_add", a, b, self.jsz)
    def add_mixed(self, a, b): return self._g2("oracle_g_add_mixed", a, b, self.jsz)
    def add_affine(self, a, b): return self._g2("oracle_g_add_affine", a, b, self.jsz)
    def double(self, a): return self._g1("oracle_g_double", a, self.jsz)
    def neg(self, a): return self._g1("oracle_g_neg", a, self.jsz)
    def normalize(self, a): return self._g1("oracle_g_normalize", a, self.jsz)
    def to_affine(self, a): return self._g1("oracle_g_to_affine", a, self.asz)
    def eq(self, a, b): return self.L.oracle_g_eq(self.cid, self.group, bytes(a), bytes(b)) == 1

    def to_jacobian(self, aff):
        if aff == bytes(self.asz):
            return self.zero()
        return bytes(aff) + self.one_field()

    def zero(self):
        return bytes(self.fsz) + self.one_field() + bytes(self.fsz)

    def one_field(self):
        one = self.to_montgomery((1).to_bytes(self.n8, "little"))
        return one + bytes(self.fsz - self.n8)

    def times_scalar(self, p, scalar):
        out = _buf(self.jsz)
        assert self.L.oracle_g_times_scalar(self.cid, self.group, bytes(p), bytes(scalar), len(scalar), out) == 0
        return out.raw

    def batch_times_scalar_affine(self, p, scalars, scalar_len, n):
        out = _buf(self.asz * n)
        assert self.L.oracle_g_batch_times_scalar_affine(self.cid, self.group, bytes(p), bytes(scalars), scalar_len, n, out) == 0
        return out.raw

    # --- batch conversions either side of the MSM (batchToAffine, batchToJacobian, batchLEMtoU, batchUtoLEM) ---
    def _batch(self, fn, data, in_sz, out_sz):
        n = len(data) // in_sz
        out = _buf(max(1, n * out_sz))
        assert getattr(self.L, fn)(self.cid, self.group, bytes(data), n, out) == 0
        return out.raw[:n * out_sz]

    def batch_to_affine(self, jac): return self._batch("oracle_g_batch_to_affine", jac, self.jsz, self.asz)
    def batch_to_jacobian(self, aff): return self._batch("oracle_g_batch_to_jacobian", aff, self.asz, self.jsz)
    def batch_lem_to_u(self, aff): return self._batch("oracle_g_batch_lem_to_u", aff, self.asz, self.asz)
    def batch_u_to_lem(self, u): return self._batch("oracle_g_batch_u_to_lem", u, self.asz, self.asz)

    def batch_lem_to_c(self, aff): return self._batch("oracle_g_batch_lem_to_c", aff, self.asz, self.fsz)

    def batch_c_to_lem(self, comp):
        n = len(comp) // self.fsz
        out = _buf(max(1, n) * self.asz)
        rc = self.L.oracle_g_batch_c_to_lem(self.cid, self.group, bytes(comp), n, out)
        if rc != 0:
            raise ValueError(f"element {rc - 1}: x^3 + b is not a square")
        return out.raw[:n * self.asz]

    def batch_in_curve_affine(self, aff):
        n = len(aff) // self.asz
        out = _buf(max(1, n))
        assert self.L.oracle_g_batch_in_curve_affine(self.cid, self.group, bytes(aff), n, out) == 0
        return list(out.raw[:n])

    def batch_in_group_affine(self, aff):
        n = len(aff) // self.asz
        out = _buf(max(1, n))
        assert self.L.oracle_g_batch_in_group_affine(self.cid, self.group, bytes(aff), n, out) == 0
        return list(out.raw[:n])

    def batch_apply_key(self, points, first, inc, jacobian_in):
        """g?m_batchApplyKey (jacobian_in) / g?m_batchApplyKeyMixed: raw Jacobian outputs"""
        isz = self.jsz if jacobian_in else self.asz
        n = len(points) // isz
        out = _buf(max(1, n) * self.jsz)
        assert self.L.oracle_g_batch_apply_key(self.cid, self.group, 1 if jacobian_in else 0, bytes(points), n, bytes(first), bytes(inc), out) == 0
        return out.raw[:n * self.jsz]

    def fr_batch_apply_key(self, elems, first, inc):
        n = len(elems) // 32
        out = _buf(max(1, n) * 32)
        assert self.L.oracle_fr_batch_apply_key(self.cid, bytes(elems), n, bytes(first), bytes(inc), out) == 0
        return out.raw[:n * 32]

    def fr_fft(self, elems, inverse=False):
        n = len(elems) // 32
        out = _buf(max(1, n) * 32)
        assert self.L.oracle_fr_fft(self.cid, 1 if inverse else 0, bytes(elems), n, out) == 0
        return out.raw[:n * 32]

    def g_fft(self, jac, inverse=False):
        n = len(jac) // self.jsz
        out = _buf(max(1, n) * self.jsz)
        assert self.L.oracle_g_fft(self.cid, self.group, 1 if inverse else 0, bytes(jac), n, out) == 0
        return out.raw[:n * self.jsz]

    # --- field ops ---
    def f_mul(self, x, y): return self._g2("oracle_f_mul", x, y, self.fsz)
    def f_add(self, x, y): return self._g2("oracle_f_add", x, y, self.fsz)
    def f_sub(self, x, y): return self._g2("oracle_f_sub", x, y, self.fsz)
    def f_square(self, x): return self._g1("oracle_f_square", x, self.fsz)
    def f_inverse(self, x): return self._g1("oracle_f_inverse", x, self.fsz)

    def f_neg(self, x): return self.f_sub(bytes(self.fsz), x)

    def to_montgomery(self, x):
        cnt = len(x) // self.n8
        out = _buf(len(x))
        assert self.L.oracle_f_to_montgomery(self.cid, bytes(x), cnt, out) == 0
        return out.raw

    def from_montgomery(self, x):
        cnt = len(x) // self.n8
        out = _buf(len(x))
        assert self.L.oracle_f_from_montgomery(self.cid, bytes(x), cnt, out) == 0
        return out.raw
