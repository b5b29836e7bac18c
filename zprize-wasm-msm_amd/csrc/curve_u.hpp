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
    // out-of-line forms; operands by reference (two large by-value structs were seen to arrive corrupted in a callee).
    // Since round 4 no shipped kernel calls them (a call keeps three points in the caller's private segment, and the HSA runtime turns
    // bytes of private segment into half a gigabyte of scratch per hardware queue: bucket_kernels.hpp, tadd); they remain for the
    // reduction built without MSM_G2_REDUCE_INLINE (experiments) and for the host cross-checks.
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

// ---- an XYZZ point of a G1 kind split over a LANE PAIR: the even lane holds (x, zz), the odd lane (y, zzz) --------------------
// Used by the bucket reduction, which is a chain of ~40 dependent full additions on few waves: one lane per point runs the 14
// field products of an addition (add-2008-s) one after the other with both operands live (256 registers).  The products of an
// addition are two nearly independent halves -- the x side (U1, U2, P, PP, PPP, Q, X3, ZZ3) and the y side (S1, S2, R, RR,
// S1*PPP, Y3, ZZZ3) -- so a pair of lanes runs them side by side: EIGHT product rounds instead of fourteen, half the live state
// per lane, four DPP exchanges (quad_perm [1,0,3,2], as the Fp2 kernels use).  Both lanes execute one instruction stream; where
// the two sides multiply different things the operands are picked by lane parity (pair_select).  Round by round (E | O):
//   m1 = a.c0*b.c1    U1 = X1 ZZ2        | S1 = Y1 ZZZ2
//   m2 = b.c0*a.c1    U2 = X2 ZZ1        | S2 = Y2 ZZZ1          d = m2 - m1:  P | R
//   sq = d^2          PP                 | RR                    (both zero <=> the same point: doubling)
//   t  = a.c1*b.c1    ZZ1 ZZ2            | ZZZ1 ZZZ2             exchange d, sq: both lanes now hold P, PP; E also RR
//   ppp = P*PP        (both lanes)
//   r6                Q = U1 PP          | ZZZ3 = t PPP
//   r7                ZZ3 = t PP         | S1 PPP                E: X3 = RR - PPP - 2Q; exchange Q - X3
//   r8                (idle)             | Y3 = R (Q - X3) - S1 PPP
// Doubling (dbl-2008-s-1) the same way in five rounds.  Case analysis as XYZZU::add (build_curve_jacobian_a0.js:583-629); every
// predicate is reduced over the pair, so both lanes always take the same path.
template <class P>
struct XYZZH {
    using K = UG1<P>;
    using C0 = typename K::L;   // x | y (a stored y, FuM<3>, fits the looser bounds of a stored x)
    using C1 = typename K::M;   // zz | zzz
    C0 c0;
    C1 c1;

    MSM_D static XYZZH inf() {
        XYZZH r;
        r.c0 = pair_parity() ? fstore<C0>(K::one()) : K::template zero<C0>();   // (x, y) = (0, 1)
        r.c1 = K::template zero<C1>();
        return r;
    }
    MSM_D bool is_inf() const {
        const uint32_t z = fis_zero(c1) ? 1u : 0u;   // zz and zzz vanish together
        return (z & pair_xchg(z)) != 0;
    }
    // this lane's half of bucket `b` as the accumulation kernels store it (XYZZU<UG1>: x, y, zz, zzz of NL words each)
    MSM_D static XYZZH load_bucket(const uint8_t* __restrict__ b) {
        constexpr int NL = UC<P>::NL;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(b) + pair_parity() * NL;
        XYZZH r;
#pragma unroll
        for (int i = 0; i < NL; i += 2) {   // NL-word coordinates are 8-byte aligned (NL even) or handled word-wise (NL odd)
            if constexpr (NL % 2 == 0) {
                const uint2 q0 = *reinterpret_cast<const uint2*>(w + i), q1 = *reinterpret_cast<const uint2*>(w + 2 * NL + i);
                r.c0.v[i] = q0.x;
                r.c0.v[i + 1] = q0.y;
                r.c1.v[i] = q1.x;
                r.c1.v[i + 1] = q1.y;
            } else {
                r.c0.v[i] = w[i];
                r.c1.v[i] = w[2 * NL + i];
                if (i + 1 < NL) {
                    r.c0.v[i + 1] = w[i + 1];
                    r.c1.v[i + 1] = w[2 * NL + i + 1];
                }
            }
        }
        return r;
    }
    MSM_D static XYZZH dbl(const XYZZH& p) {
        if (p.is_inf()) return p;
        const bool odd = pair_parity() != 0;
        auto u = fdbl(p.c0);                                        // O: U = 2y
        auto base = pair_select(p.c0, u, odd);                      // E: x | O: U
        auto s1 = fsqr(base);                                       // E: XX | O: V
        auto os1 = pair_xchg(s1);                                   // E: V | O: XX
        auto vv = pair_select(os1, s1, odd);                        // V on both lanes
        auto xx = pair_select(s1, os1, odd);                        // XX on both lanes
        auto r2 = fmul(base, vv);                                   // E: S = x V | O: W = U V
        auto Mm = fprep(fadd(fdbl(xx), xx));                        // M = 3 XX on both lanes
        auto r3 = fmul(pair_select(Mm, r2, odd), pair_select(Mm, p.c1, odd));   // E: M^2 | O: ZZZ3 = W zzz
        auto x3 = fnorm(fsub(r3, fdbl(r2)));                        // E: X3 = M^2 - 2S
        auto r4 = fmul(pair_select(vv, r2, odd), pair_select(p.c1, p.c0, odd)); // E: ZZ3 = V zz | O: W y
        auto ow = pair_xchg(fnorm(fsub(r2, x3)));                   // O: S - X3
        auto y3 = fnorm(fsub(fmul(Mm, ow), r4));                    // O: Y3 = M (S - X3) - W y
        XYZZH r;
        r.c0 = fstore<C0>(pair_select(x3, y3, odd));
        r.c1 = fstore<C1>(pair_select(r4, r3, odd));
        return r;
    }
    MSM_D static XYZZH add(const XYZZH& a, const XYZZH& b) {
        if (a.is_inf()) return b;
        if (b.is_inf()) return a;
        const bool odd = pair_parity() != 0;
        auto m1 = fmul(a.c0, b.c1);                                 // E: U1 | O: S1
        auto m2 = fmul(b.c0, a.c1);                                 // E: U2 | O: S2
        auto d = fprep(fsub(m2, m1));                               // E: P  | O: R
        auto sq = fsqr(d);                                          // E: PP | O: RR
        {
            const uint32_t z = fis_zero(sq) ? 1u : 0u;
            if (z & pair_xchg(z)) return dbl(a);                    // P = R = 0: the same point
        }
        auto t = fmul(a.c1, b.c1);                                  // E: ZZ1 ZZ2 | O: ZZZ1 ZZZ2
        auto od = pair_xchg(d);
        auto osq = pair_xchg(sq);                                   // E: RR | O: PP
        auto pd = pair_select(d, od, odd);                          // P on both lanes
        auto pp = pair_select(sq, osq, odd);                        // PP on both lanes
        auto ppp = fmul(pd, pp);                                    // PPP on both lanes
        auto r6 = fmul(pair_select(m1, t, odd), pair_select(pp, ppp, odd));   // E: Q = U1 PP | O: ZZZ3 = t PPP
        auto r7 = fmul(pair_select(t, m1, odd), pair_select(pp, ppp, odd));   // E: ZZ3 = t PP | O: S1 PPP
        auto x3 = fnorm(fsub(osq, fadd(ppp, fdbl(r6))));            // E: X3 = RR - PPP - 2Q
        auto ow = pair_xchg(fnorm(fsub(r6, x3)));                   // O: Q - X3
        auto y3 = fnorm(fsub(fmul(d, ow), r7));                     // O: Y3 = R (Q - X3) - S1 PPP
        XYZZH r;
        r.c0 = fstore<C0>(pair_select(x3, y3, odd));
        r.c1 = fstore<C1>(pair_select(r7, r6, odd));
        return r;
    }
    // canonical packed image XYZZ<Packed>: this lane writes its two coordinates
    MSM_D void to_packed(uint32_t* w) const {
        const uint32_t par = pair_parity();
        K::store(w + par * K::WORDS, c0);
        K::store(w + (2 + par) * K::WORDS, c1);
    }
};

#endif  // __HIPCC__

}  // namespace msm
