// curve.hpp -- group law for y^2 = x^3 + b (a = 0) used by the MSM engine, generic over the
// coordinate field F (Fp for G1, Fp2 for G2), host and device.
//
// Replaces the reference's group layer on the hot path
//   wasmcurves/src/build_curve_jacobian_a0.js  (_addMixed :661-761, _add :541-658,
//   _double :291-359, _doubleAffine :362-428, _normalize :940-973, _toAffine :977-1011,
//   zero/isZero conventions :40-171).
// The reference accumulates buckets in Jacobian coordinates (7M+4S mixed add).  Here buckets are
// kept in extended Jacobian ("XYZZ": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), whose mixed add is
// 8M+2S with no field inversion and whose exceptional cases reduce to one doubling branch.
// Every exceptional case (infinity operands, P+P, P+(-P)) yields the same group element as the
// reference's branches; results leave the engine as normalised affine/Jacobian bytes, so the
// internal coordinate system is not observable.
//
// Infinity conventions at the boundary (build_curve_jacobian_a0.js:40-77,124-171):
//   affine infinity  <=> x == 0 && y == 0;   Jacobian infinity <=> Z == 0, written as (0, R, 0).
#pragma once
#include "field.hpp"

namespace msm {

template <class F>
struct Affine {
    F x, y;
    MSM_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    MSM_HD static Affine canon(const Affine& p) { return {F::canon(p.x), F::canon(p.y)}; }
};

template <class F>
struct Jacobian {
    F x, y, z;
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;

    MSM_HD static XYZZ inf() { return {F::zero(), F::one(), F::zero(), F::zero()}; }
    MSM_HD bool is_inf() const { return zz.is_zero(); }
    MSM_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        return {p.x, p.y, F::one(), F::one()};
    }

    // 2*(x,y) for an affine, non-infinite point (mdbl-2008-s-1, a = 0).  Rare on the hot path (only
    // when a bucket meets the same point twice in a row), so kept out of line on the device.
#if defined(__HIP_DEVICE_COMPILE__)
    // by-value operands of an out-of-line device function: two 384-byte structs passed this way arrived corrupted in the callee
    // (DESIGN.md section 7); anything wider than MSM_BYVALUE_MAX bytes must travel by reference (add_to / dbl_to below)
    __device__ __noinline__ static XYZZ dbl_affine(F x, F y) {
        static_assert(2 * sizeof(F) <= MSM_BYVALUE_MAX, "pass wide operands of out-of-line device functions by reference");
        return dbl_affine_inl(x, y);
    }
#else
    static XYZZ dbl_affine(const F& x, const F& y) { return dbl_affine_inl(x, y); }
#endif
    MSM_HD static XYZZ dbl_affine_inl(const F& x, const F& y) {
        F U = F::dbl(y);
        F V = F::sqr(U);
        F W = F::mul(U, V);
        F S = F::mul(x, V);
        F xx = F::sqr(x);
        F M = F::add(F::dbl(xx), xx);
        F X3 = F::sub(F::sqr(M), F::dbl(S));
        F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, y));
        return {X3, Y3, V, W};
    }
    // 2*p (dbl-2008-s-1, a = 0); y == 0 cannot occur for points of odd order, and gives ZZ = 0
    MSM_HD static XYZZ dbl(const XYZZ& p) {
        if (p.is_inf()) return p;
        F U = F::dbl(p.y);
        F V = F::sqr(U);
        F W = F::mul(U, V);
        F S = F::mul(p.x, V);
        F xx = F::sqr(p.x);
        F M = F::add(F::dbl(xx), xx);
        F X3 = F::sub(F::sqr(M), F::dbl(S));
        F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
        return {X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
    }

    // acc += (x2, y2) affine (madd-2008-s), 8M + 2S.  Same case analysis as
    // build_curve_jacobian_a0.js:697-733: acc infinite -> copy; point infinite -> unchanged;
    // equal points -> doubling; opposite points -> infinity (falls out of the formula: PP = 0).
    MSM_HD static void madd(XYZZ& acc, const F& x2, const F& y2) {
        if (x2.is_zero() && y2.is_zero()) return;
        if (acc.is_inf()) {
            acc = {x2, y2, F::one(), F::one()};
            return;
        }
        F U2 = F::mul(x2, acc.zz);
        F S2 = F::mul(y2, acc.zzz);
        F Pd = F::sub(U2, acc.x);
        F Rd = F::sub(S2, acc.y);
        if (Pd.is_zero() && Rd.is_zero()) {
            acc = dbl_affine(x2, y2);
            return;
        }
        F PP = F::sqr(Pd);
        F PPP = F::mul(Pd, PP);
        F Q = F::mul(acc.x, PP);
        F X3 = F::sub(F::sub(F::sub(F::sqr(Rd), PPP), Q), Q);
        F Y3 = F::sub(F::mul(Rd, F::sub(Q, X3)), F::mul(acc.y, PPP));
        acc.x = X3;
        acc.y = Y3;
        acc.zz = F::mul(acc.zz, PP);
        acc.zzz = F::mul(acc.zzz, PPP);
    }

    // a + b (add-2008-s), 12M + 2S, with the case analysis of build_curve_jacobian_a0.js:583-629
    MSM_HD static XYZZ add(const XYZZ& a, const XYZZ& b) {
        if (a.is_inf()) return b;
        if (b.is_inf()) return a;
        F U1 = F::mul(a.x, b.zz);
        F U2 = F::mul(b.x, a.zz);
        F S1 = F::mul(a.y, b.zzz);
        F S2 = F::mul(b.y, a.zzz);
        F Pd = F::sub(U2, U1);
        F Rd = F::sub(S2, S1);
        if (Pd.is_zero() && Rd.is_zero()) return dbl(a);
        F PP = F::sqr(Pd);
        F PPP = F::mul(Pd, PP);
        F Q = F::mul(U1, PP);
        F X3 = F::sub(F::sub(F::sub(F::sqr(Rd), PPP), Q), Q);
        F Y3 = F::sub(F::mul(Rd, F::sub(Q, X3)), F::mul(S1, PPP));
        F ZZ3 = F::mul(F::mul(a.zz, b.zz), PP);
        F ZZZ3 = F::mul(F::mul(a.zzz, b.zzz), PPP);
        return {X3, Y3, ZZ3, ZZZ3};
    }

    // out-of-line forms for kernels that chain many full additions (bucket reduction)
    // operands by reference: they stay in the caller's frame (two 384-byte by-value arguments of the BLS12-381 G2 point were seen
    // to arrive corrupted in the FFT kernels)
    __device__ __noinline__ static void add_to(XYZZ& r, const XYZZ& a, const XYZZ& b) { r = add(a, b); }
    __device__ __noinline__ static void dbl_to(XYZZ& r, const XYZZ& a) { r = dbl(a); }
    MSM_D static XYZZ add_ni(const XYZZ& a, const XYZZ& b) {
        XYZZ r;
        add_to(r, a, b);
        return r;
    }
    MSM_D static XYZZ dbl_ni(const XYZZ& a) {
        XYZZ r;
        dbl_to(r, a);
        return r;
    }

    // canonical representation of every coordinate (device values are relaxed, see field.hpp)
    MSM_HD static XYZZ canon(const XYZZ& p) { return {F::canon(p.x), F::canon(p.y), F::canon(p.zz), F::canon(p.zzz)}; }

    MSM_HD static XYZZ neg(const XYZZ& p) { return {p.x, F::neg(p.y), p.zz, p.zzz}; }

    // Jacobian (X,Y,Z) -> XYZZ: (X, Y, Z^2, Z^3); Z == 0 -> infinity.  (0, 0, Z) is also infinity:
    // the reference's addMixed(infinity, affine infinity) writes (0, 0, R) (build_curve_jacobian_a0.js
    // :697-704) and every later reference operation treats that triple as the identity.
    MSM_HD static XYZZ from_jacobian(const Jacobian<F>& p) {
        if (p.z.is_zero() || (p.x.is_zero() && p.y.is_zero())) return inf();
        F zz = F::sqr(p.z);
        return {p.x, p.y, zz, F::mul(zz, p.z)};
    }
    // -> affine (x, y); infinity -> (0, 0) as build_curve_jacobian_a0.js:977-1011
    MSM_HD static Affine<F> to_affine(const XYZZ& p) {
        if (p.is_inf()) return {F::zero(), F::zero()};
        // 1/ZZ = ZZ^2 * ZZZ^-2 ... cheaper: one inversion of ZZ*ZZZ
        F t = F::inv(F::mul(p.zz, p.zzz));
        F izz = F::mul(t, p.zzz);
        F izzz = F::mul(t, p.zz);
        return {F::mul(p.x, izz), F::mul(p.y, izzz)};
    }
};

}  // namespace msm
