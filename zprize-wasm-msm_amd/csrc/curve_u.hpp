// curve_u.hpp -- the group law of curve.hpp (XYZZ coordinates, same case analysis, same citations:
// wasmcurves/src/build_curve_jacobian_a0.js _addMixed :661-761, _add :541-658, _double :291-359,
// _doubleAffine :362-428) over the unsaturated field representation of field_u.hpp.  Device only; used by
// the bucket kernels.  Buckets live in HBM in this representation; everything that crosses the engine's
// boundary is converted back to the packed canonical image (to_packed).
//
// The formulas are written once, generically over a "kind" K:
//   UG1<P>: coordinates in Fp  (Fu),   UG2<P>: coordinates in Fp2 (Fp2u).
// Bounds of intermediate values are tracked by the types (see field_u.hpp); the stored coordinate types
// K::L / K::Y / K::M are the loop invariants of the accumulation and fstore<> static_asserts that every
// value written back respects them.
#pragma once
#include "curve.hpp"
#include "field_u.hpp"

namespace msm {

#if defined(__HIPCC__)

// ---- small generic helpers over Fu / Fp2u -------------------------------------------------
// Write-back into a stored coordinate type.  A value whose bounds exceed the type's is first brought down:
// limbs by a carry pass, the value itself by a Montgomery product with one.  Both are decided at compile time;
// for BLS12-381 and for BN254 G1 neither is ever inserted on the accumulation path (the declared invariants
// close), BN254 G2 pays a product here and there because R'/q is only 169.
template <class T, class P, uint32_t L, uint32_t V>
MSM_HD T fstore(const Fu<P, L, V>& a) {
    if constexpr (V > T::VB) {
        FuM<P, 1> one;
#pragma unroll
        for (int i = 0; i < UC<P>::NL; ++i) one.v[i] = P::UONE[i];
        return fstore<T>(fmul(a, one));
    } else if constexpr (L > T::LMAX) {
        return fstore<T>(fnorm(a));
    } else {
        return uwiden<T>(a);
    }
}
template <class T, class A>
MSM_HD T fstore(const Fp2u<A>& a) {
    return {fstore<typename T::Base>(a.c0), fstore<typename T::Base>(a.c1)};
}
// a*b - c*d
template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB, uint32_t LC, uint32_t VC, uint32_t LD, uint32_t VD>
MSM_HD auto fmulsub(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b, const Fu<P, LC, VC>& c, const Fu<P, LD, VD>& d) {
    return fmul2(a, b, c, fneg(d));   // one reduction for both products
}
template <class A, class B, class C_, class D>
MSM_HD auto fmulsub(const Fp2u<A>& a, const Fp2u<B>& b, const Fp2u<C_>& c, const Fp2u<D>& d) {
    return fnorm(fsub(fmul(a, b), fmul(c, d)));
}
template <class T, class A>
MSM_HD T fstore(const Fp2d<A>& a) {
    return {fstore<typename T::Base>(a.v)};
}
template <class A, class B, class C_, class D>
MSM_HD auto fmulsub(const Fp2d<A>& a, const Fp2d<B>& b, const Fp2d<C_>& c, const Fp2d<D>& d) {
    return fnorm(fsub(fmul(a, b), fmul(c, d)));
}
// normalise once a value that is about to be used as a multiplier operand several times
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fprep(const Fu<P, LA, VA>& a) {
    if constexpr (UC<P>::cols_fit(LA, LA, 0, 0)) return a;
    else return fnorm(a);
}
template <class A>
MSM_HD auto fprep(const Fp2u<A>& a) {
    using C = UC<typename A::Params>;
    if constexpr (C::cols_fit(A::LMAX, A::LMAX, A::LMAX, C::sub_cmax(A::LMAX))) return a;
    else return fnorm(a);
}

template <class A>
MSM_HD auto fprep(const Fp2d<A>& a) {
    using C = UC<typename A::Params>;
    if constexpr (C::cols_fit(A::LMAX, A::LMAX, A::LMAX, C::sub_cmax(A::LMAX))) return a;
    else return fnorm(a);
}

// ---- kinds ---------------------------------------------------------------------------------
template <class P>
struct UG1 {
    using Params = P;
    using Packed = Fp<P>;
    using In = FuIn<P>;                        // a packed coordinate as loaded
    using L = Fu<P, UC<P>::MASK + 31, 16>;     // stored x: a normalised lazy difference
    using Y = FuM<P, 3>;                       // stored y: always a Montgomery output
    using M = FuM<P, 3>;                       // stored zz, zzz
    static constexpr int WORDS = P::N;         // packed words per coordinate
    static constexpr int LANES = 1;            // lanes that share one point
    static constexpr bool INLINE_RARE = true;
    MSM_HD static In load(const uint32_t* w) { return u_from_packed<P>(w); }
    template <class T>
    MSM_HD static void store(uint32_t* w, const T& a) { u_to_packed(w, a); }
    MSM_HD static FuM<P, 1> one() {
        FuM<P, 1> r;
#pragma unroll
        for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = P::UONE[i];
        return r;
    }
    template <class T>
    MSM_HD static T zero() {
        T r;
#pragma unroll
        for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = 0;
        return r;
    }
};
template <class P>
struct UG2 {
    using Params = P;
    using Packed = Fp2<P>;
    using In = Fp2u<FuIn<P>>;
    using L = Fp2u<Fu<P, UC<P>::MASK + 31, 16>>;
    using Y = Fp2u<Fu<P, UC<P>::MASK + 31, 16>>;   // a normalised lazy difference of two products
    using M = Fp2u<FuM<P, 3>>;
    static constexpr int WORDS = 2 * P::N;
    static constexpr int LANES = 1;
    static constexpr bool INLINE_RARE = false;
    MSM_HD static In load(const uint32_t* w) { return {u_from_packed<P>(w), u_from_packed<P>(w + P::N)}; }
    template <class T>
    MSM_HD static void store(uint32_t* w, const T& a) {
        u_to_packed(w, a.c0);
        u_to_packed(w + P::N, a.c1);
    }
    MSM_HD static Fp2u<FuM<P, 1>> one() { return {UG1<P>::one(), UG1<P>::template zero<FuM<P, 1>>()}; }
    template <class T>
    MSM_HD static T zero() {
        return {UG1<P>::template zero<typename T::Base>(), UG1<P>::template zero<typename T::Base>()};
    }
};
// Fp2 coordinates distributed over lane pairs (field_u.hpp, Fp2d): each lane holds one component of every coordinate.
template <class P>
struct UG2P {
    using Params = P;
    using Packed = Fp2<P>;
    using In = Fp2d<FuIn<P>>;
    using L = Fp2d<Fu<P, UC<P>::MASK + 31, 16>>;
    using Y = Fp2d<Fu<P, UC<P>::MASK + 31, 16>>;
    using M = Fp2d<FuM<P, 3>>;
    static constexpr int WORDS = 2 * P::N;     // packed words per coordinate (both components)
    static constexpr int LANES = 2;
    static constexpr bool INLINE_RARE = true;
    // w: the packed coordinate c0 || c1; this lane takes its own component
    MSM_HD static In load(const uint32_t* w) { return {u_from_packed<P>(w + pair_parity() * P::N)}; }
    template <class T>
    MSM_HD static void store(uint32_t* w, const T& a) { u_to_packed(w + pair_parity() * P::N, a.v); }
    MSM_HD static Fp2d<FuM<P, 1>> one() {
        Fp2d<FuM<P, 1>> r;
        const bool odd = pair_parity() != 0;
#pragma unroll
        for (int i = 0; i < UC<P>::NL; ++i) r.v.v[i] = odd ? 0u : P::UONE[i];
        return r;
    }
    template <class T>
    MSM_HD static T zero() {
        return {UG1<P>::template zero<typename T::Base>()};
    }
};
// which kind serves a packed coordinate field
template <class F>
struct UKind;
template <class P, bool INL>
struct UKind<Fp<P, INL>> {
    using type = UG1<P>;
};
#ifndef MSM_G2_PAIRED
#define MSM_G2_PAIRED 1   // 0: one lane per Fp2 point (UG2), base-field products out of line
#endif
template <class B>
struct UKind<Fp2T<B>> {
#if MSM_G2_PAIRED
    using type = UG2P<typename B::Params>;
#else
    using type = UG2<typename B::Params>;
#endif
};

MSM_HD bool words_all_zero(const uint32_t* w, int n) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < n; ++i) o |= w[i];
    return o == 0;
}

// ---- XYZZ point in the unsaturated representation --------------------------------------------
template <class K>
struct XYZZU {
    typename K::L x;
    typename K::Y y;
    typename K::M zz, zzz;

    MSM_HD static XYZZU inf() {
        XYZZU r;
        r.x = K::template zero<typename K::L>();
        r.y = fstore<typename K::Y>(K::one());
        r.zz = K::template zero<typename K::M>();
        r.zzz = K::template zero<typename K::M>();
        return r;
    }
    MSM_HD bool is_inf() const { return fis_zero(zz); }

    // 2*(x, y), affine and finite (mdbl-2008-s-1).  Rare on the hot path.  G1 expands it inline: a callee's register
    // allocation is not governed by the kernel's occupancy target, and as a call it pinned k_accumulate at 200 VGPRs
    // (2 waves/SIMD); inline the kernel needs 165 (3 waves).  G2 keeps it out of line (code size).
    template <class X2, class Y2>
    __host__ __device__ __noinline__ static XYZZU dbl_affine(X2 x_in, Y2 y_in) {
        static_assert(sizeof(X2) + sizeof(Y2) <= MSM_BYVALUE_MAX, "pass wide operands of out-of-line device functions by reference");
        return dbl_affine_inl(x_in, y_in);
    }
    template <class X2, class Y2>
    MSM_HD static XYZZU dbl_affine_inl(const X2& x_in, const Y2& y_in) {
        auto one = K::one();
        auto x = fmul(x_in, one);   // brings the value bound of a loaded coordinate down to a product's
        auto y = fmul(y_in, one);
        auto U = fdbl(y);
        auto V = fsqr(U);
        auto Wv = fmul(U, V);
        auto S = fmul(x, V);
        auto xx = fsqr(x);
        auto Mm = fprep(fadd(fdbl(xx), xx));
        auto X3 = fnorm(fsub(fsqr(Mm), fdbl(S)));
        auto Y3 = fmulsub(Mm, fsub(S, X3), Wv, y);
        XYZZU r;
        r.x = fstore<typename K::L>(X3);
        r.y = fstore<typename K::Y>(Y3);
        r.zz = fstore<typename K::M>(V);
        r.zzz = fstore<typename K::M>(Wv);
        return r;
    }

    // acc += (x2, y2) affine (madd-2008-s, 8M + 2S); y2 may already carry the digit's sign.
    // Case analysis as build_curve_jacobian_a0.js:697-733: acc infinite -> copy; equal points -> doubling;
    // opposite points -> infinity (falls out of the formula: PP = 0 makes ZZ3 = 0).
    // The caller has already skipped affine infinity (x2 == y2 == 0 in the packed image).
    template <class X2, class Y2>
    MSM_HD static void madd(XYZZU& acc, const X2& x2, const Y2& y2) {
        if (acc.is_inf()) {
            auto one = K::one();
            acc.x = fstore<typename K::L>(fmul(x2, one));
            acc.y = fstore<typename K::Y>(fmul(y2, one));
            acc.zz = fstore<typename K::M>(one);
            acc.zzz = fstore<typename K::M>(one);
            return;
        }
        auto U2 = fmul(x2, acc.zz);
        auto S2 = fmul(y2, acc.zzz);
        auto Pd = fprep(fsub(U2, acc.x));
        auto Rd = fprep(fsub(S2, acc.y));
        auto PP = fsqr(Pd);
        auto RR = fsqr(Rd);
        if (fis_zero(PP) && fis_zero(RR)) {   // same point: P = R = 0
            if constexpr (K::INLINE_RARE) acc = dbl_affine_inl(x2, y2);
            else acc = dbl_affine(x2, y2);
            return;
        }
        auto PPP = fmul(Pd, PP);
        auto Q = fmul(acc.x, PP);
        auto X3 = fnorm(fsub(RR, fadd(PPP, fdbl(Q))));
        auto Y3 = fmulsub(Rd, fsub(Q, X3), PPP, acc.y);
        acc.x = fstore<typename K::L>(X3);
        acc.y = fstore<typename K::Y>(Y3);
        acc.zz = fstore<typename K::M>(fmul(acc.zz, PP));
        acc.zzz = fstore<typename K::M>(fmul(acc.zzz, PPP));
    }

    // 2*p (dbl-2008-s-1)
    MSM_HD static XYZZU dbl(const XYZZU& p) {
        if (p.is_inf()) return p;
        auto U = fdbl(p.y);
        auto V = fsqr(U);
        auto Wv = fmul(U, V);
        auto S = fmul(p.x, V);
        auto xx = fsqr(p.x);
        auto Mm = fprep(fadd(fdbl(xx), xx));
        auto X3 = fnorm(fsub(fsqr(Mm), fdbl(S)));
        auto Y3 = fmulsub(Mm, fsub(S, X3), Wv, p.y);
        XYZZU r;
        r.x = fstore<typename K::L>(X3);
        r.y = fstore<typename K::Y>(Y3);
        r.zz = fstore<typename K::M>(fmul(V, p.zz));
        r.zzz = fstore<typename K::M>(fmul(Wv, p.zzz));
        return r;
    }

    // a + b (add-2008-s, 12M + 2S), case analysis of build_curve_jacobian_a0.js:583-629
    MSM_HD static XYZZU add(const XYZZU& a, const XYZZU& b) {
        if (a.is_inf()) return b;
        if (b.is_inf()) return a;
        auto U1 = fmul(a.x, b.zz);
        auto U2 = fmul(b.x, a.zz);
        auto S1 = fmul(a.y, b.zzz);
        auto S2 = fmul(b.y, a.zzz);
        auto Pd = fprep(fsub(U2, U1));
        auto Rd = fprep(fsub(S2, S1));
        auto PP = fsqr(Pd);
        auto RR = fsqr(Rd);
        if (fis_zero(PP) && fis_zero(RR)) return dbl(a);
        auto PPP = fmul(Pd, PP);
        auto Q = fmul(U1, PP);
        auto X3 = fnorm(fsub(RR, fadd(PPP, fdbl(Q))));
        auto Y3 = fmulsub(Rd, fsub(Q, X3), S1, PPP);
        XYZZU r;
        r.x = fstore<typename K::L>(X3);
        r.y = fstore<typename K::Y>(Y3);
        r.zz = fstore<typename K::M>(fmul(fmul(a.zz, b.zz), PP));
        r.zzz = fstore<typename K::M>(fmul(fmul(a.zzz, b.zzz), PPP));
        return r;
    }
    // out-of-line forms; operands by reference (two large by-value structs were seen to arrive corrupted in a callee)
    __host__ __device__ __noinline__ static void add_to(XYZZU& r, const XYZZU& a, const XYZZU& b) { r = add(a, b); }
    __host__ __device__ __noinline__ static void dbl_to(XYZZU& r, const XYZZU& a) { r = dbl(a); }
    MSM_HD static XYZZU add_ni(const XYZZU& a, const XYZZU& b) {
        XYZZU r;
        add_to(r, a, b);
        return r;
    }
    MSM_HD static XYZZU dbl_ni(const XYZZU& a) {
        XYZZU r;
        dbl_to(r, a);
        return r;
    }

    // canonical packed image XYZZ<Packed> (what the host combine reads)
    MSM_HD void to_packed(uint32_t* w) const {
        K::store(w, x);
        K::store(w + K::WORDS, y);
        K::store(w + 2 * K::WORDS, zz);
        K::store(w + 3 * K::WORDS, zzz);
    }
};

#endif  // __HIPCC__

}  // namespace msm
