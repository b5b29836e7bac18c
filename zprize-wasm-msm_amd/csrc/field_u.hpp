// field_u.hpp -- "unsaturated" Montgomery arithmetic for the bucket kernels (device only).
//
// Same field as field.hpp (wasmcurves/src/build_f1m.js f1m_mul :466-777, f1m_square :779-1076,
// f1m_add/sub/neg :71-117; build_f2m.js f2m_mul :152-188, f2m_square :211-252), different machine
// representation.  An element is UN limbs of UW bits held in 32-bit words (BLS12-381: 14 x 28,
// BN254: 9 x 29), Montgomery radix R' = 2^(UW*UN).  Why: on gfx950 v_mad_u64_u32 (32x32 + 64 -> 64) is a
// full-rate instruction, and with limbs this narrow a whole product column -- UN partial products of the
// operands plus UN of the reduction -- fits a 64-bit accumulator.  A partial product then costs exactly ONE
// instruction; the saturated 32-bit-limb multiplier of field.hpp pays a second one (v_addc_co_u32) per
// product to catch the carry.  Additions and subtractions are limb-wise with no carry chain and no
// conditional correction: values simply grow ("lazy") and are renormalised only when a multiplier input
// would overflow a column.
//
// Correctness is carried by the TYPES.  Fu<P, LMAX, VB> is an element whose every limb is <= LMAX and whose
// value is < VB*q.  Every operation computes the bounds of its result at compile time; the multiplier
// static_asserts that its columns fit 64 bits (normalising an operand first when they would not), the
// subtraction picks a multiple of q whose limbs dominate the subtrahend's, and the stored coordinate types
// of curve_u.hpp static_assert that what is written back respects their declared bounds.  No bound is
// argued in comments only.
//
// Boundary: packed elements (the reference's byte image, N 32-bit limbs, radix R = 2^(32N)) become
// x*R' by re-slicing the SAME bits shifted left by USHIFT = UW*UN - 32N -- no multiplication on the way in
// (the shifted value is a perfectly good lazy element); on the way out one Montgomery product by (R mod q) and a final conditional subtraction restore the
// canonical packed image.
#pragma once
#include "field.hpp"

namespace msm {

template <class P>
struct UC {
    static constexpr int W = P::UW;
    static constexpr int NL = P::UN;
    static constexpr uint32_t MASK = (1u << W) - 1;
    static constexpr uint32_t RQ = P::URQ;   // floor(R'/q): a value < VB*q fits the limbs while VB <= RQ + 1
    struct Limbs {
        uint32_t v[NL];
    };
    // m*q, lower limbs normalised
    static constexpr Limbs mulq(uint32_t m) {
        Limbs r{};
        uint64_t c = 0;
        for (int i = 0; i < NL; ++i) {
            uint64_t t = (uint64_t)P::UQ[i] * m + c;
            if (i < NL - 1) {
                r.v[i] = (uint32_t)t & MASK;
                c = t >> W;
            } else {
                r.v[i] = (uint32_t)t;
            }
        }
        return r;
    }
    // largest top limb an element of value < vb*q and limbs <= lmax can have
    static constexpr uint64_t top_bound(uint32_t lmax, uint32_t vb) {
        uint64_t t = (uint64_t)vb * ((uint64_t)P::UQ[NL - 1] + 1);
        return t < lmax ? t : lmax;
    }
    // subtraction a - b is computed as a + (C - b) with C = m*q re-spread so that every limb of C dominates
    // the matching limb of b:  C_i = (m*q)_i + K*2^W - (i > 0 ? K : 0),  C_top = (m*q)_top - K.
    static constexpr uint32_t sub_k(uint32_t lmaxb) { return (uint32_t)(((uint64_t)lmaxb + MASK - 1) / MASK); }
    static constexpr uint32_t sub_m(uint32_t lmaxb, uint32_t vbb) {
        uint64_t need = top_bound(lmaxb, vbb) + sub_k(lmaxb);
        uint32_t m = 1;
        while ((uint64_t)mulq(m).v[NL - 1] < need) ++m;
        return m;
    }
    static constexpr Limbs sub_c(uint32_t lmaxb, uint32_t vbb) {
        uint32_t K = sub_k(lmaxb);
        Limbs d = mulq(sub_m(lmaxb, vbb));
        for (int i = 0; i < NL; ++i) {
            if (i < NL - 1) d.v[i] += K << W;
            if (i > 0) d.v[i] -= K;
        }
        return d;
    }
    static constexpr uint64_t sub_cmax(uint32_t lmaxb) { return (uint64_t)MASK + ((uint64_t)sub_k(lmaxb) << W); }
    // value bound (in units of q) of a Montgomery product whose inputs satisfy value(T) < t*q*q
    static constexpr uint32_t mul_vb(uint64_t t) { return (uint32_t)(t / RQ) + 2; }
    // do all columns of  sum a_i b_j (+ sum c_i d_j) + sum m_i q_j + carry  fit 64 bits?
    static constexpr bool cols_fit(uint64_t la, uint64_t lb, uint64_t lc, uint64_t ld) {
        unsigned __int128 s = (unsigned __int128)NL * la * lb + (unsigned __int128)NL * lc * ld + (unsigned __int128)NL * MASK * MASK +
                              ((unsigned __int128)1 << (64 - W));
        return s < ((unsigned __int128)1 << 64);
    }
};

template <class P, uint32_t LMAX_, uint32_t VB_>
struct Fu {
    using Params = P;
    using C = UC<P>;
    static constexpr uint32_t LMAX = LMAX_;
    static constexpr uint32_t VB = VB_;
    static constexpr int NL = C::NL;
    static_assert(VB_ >= 1 && VB_ <= C::RQ + 1, "value bound does not fit R'");
    uint32_t v[NL];
};

// what a Montgomery product returns: limbs strictly normalised, value < V*q with V from the operands' bounds
template <class P, uint32_t V>
using FuM = Fu<P, UC<P>::MASK, V>;

#if defined(__HIPCC__)

// re-type to looser bounds (free)
template <class T, class P, uint32_t L, uint32_t V>
MSM_HD T uwiden(const Fu<P, L, V>& a) {
    static_assert(L <= T::LMAX && V <= T::VB, "uwiden would tighten a bound");
    T r;
#pragma unroll
    for (int i = 0; i < T::NL; ++i) r.v[i] = a.v[i];
    return r;
}

template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB>
MSM_HD auto fadd(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b) {
    static_assert((uint64_t)LA + LB < (1ull << 32), "limb overflow in add");
    Fu<P, LA + LB, VA + VB> r;
#pragma unroll
    for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fdbl(const Fu<P, LA, VA>& a) {
    static_assert(2ull * LA < (1ull << 32), "limb overflow in dbl");
    Fu<P, 2 * LA, 2 * VA> r;
#pragma unroll
    for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = a.v[i] << 1;
    return r;
}

template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB>
MSM_HD auto fsub(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b) {
    using C = UC<P>;
    static_assert((uint64_t)LA + C::sub_cmax(LB) < (1ull << 32), "limb overflow in sub");
    constexpr typename C::Limbs c = C::sub_c(LB, VB);
    Fu<P, LA + (uint32_t)C::sub_cmax(LB), VA + C::sub_m(LB, VB)> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = a.v[i] + (c.v[i] - b.v[i]);
    return r;
}
// -a
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fneg(const Fu<P, LA, VA>& a) {
    using C = UC<P>;
    constexpr typename C::Limbs c = C::sub_c(LA, VA);
    Fu<P, (uint32_t)C::sub_cmax(LA), C::sub_m(LA, VA)> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = c.v[i] - a.v[i];
    return r;
}
// flag ? -a : a
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fcneg(const Fu<P, LA, VA>& a, bool flag) {
    using C = UC<P>;
    constexpr typename C::Limbs c = C::sub_c(LA, VA);
    constexpr uint32_t L = (uint32_t)C::sub_cmax(LA) > LA ? (uint32_t)C::sub_cmax(LA) : LA;
    constexpr uint32_t V = C::sub_m(LA, VA) > VA ? C::sub_m(LA, VA) : VA;
    Fu<P, L, V> r;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) r.v[i] = flag ? c.v[i] - a.v[i] : a.v[i];
    return r;
}

// one parallel carry pass: limbs come back <= MASK + (LMAX >> W)
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fnorm(const Fu<P, LA, VA>& a) {
    using C = UC<P>;
    constexpr int NL = C::NL;
    Fu<P, C::MASK + (LA >> C::W) + (VA > C::RQ ? 1u : 0u), VA> r;
    r.v[0] = a.v[0] & C::MASK;
#pragma unroll
    for (int i = 1; i < NL - 1; ++i) r.v[i] = (a.v[i] & C::MASK) + (a.v[i - 1] >> C::W);
    r.v[NL - 1] = a.v[NL - 1] + (a.v[NL - 2] >> C::W);   // value < R' keeps the top limb below 2^W
    return r;
}

// ------------------------------------------------------------------------------------------
// Montgomery product(s):  r = (a*b [+ c*d]) / R'  mod q,   r < (T/(R' q) + 1) q, limbs strictly normalised.
// Product scanning: one 64-bit accumulator, one v_mad_u64_u32 per partial product.
// ------------------------------------------------------------------------------------------
// acc += a*b.  Plain C lets the compiler split a column into two interleaved chains (good for a kernel that runs one wave
// per SIMD: the bucket reduction) at the price of one 64-bit add per column to join them.  A translation unit that defines
// MSM_U_CHAINED_MADS (the G1 accumulation kernels: 3-4 waves per SIMD hide the dependent-issue latency) gets ONE chain per
// multiplication instead -- the previous column's carry is simply the addend of the next column's first product: 482 instead
// of 518 VALU instructions per 14-limb product.
MSM_HD void umad(uint64_t& acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(MSM_U_CHAINED_MADS)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
    acc += (uint64_t)a * b;
#endif
}
// same with b a wave-uniform constant (modulus limb)
MSM_HD void umad_c(uint64_t& acc, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(MSM_U_CHAINED_MADS)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(b) : "vcc");
#else
    acc += (uint64_t)a * b;
#endif
}

template <class P, bool TWO>
MSM_HD void umont(uint32_t* __restrict__ r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
    using C = UC<P>;
    constexpr int NL = C::NL;
    constexpr int W = C::W;
    uint32_t m[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            umad(acc, a[i], b[k - i]);
            if constexpr (TWO) umad(acc, c[i], d[k - i]);
        }
#pragma unroll
        for (int i = 0; i < k; ++i) umad_c(acc, m[i], P::UQ[k - i]);
        m[k] = ((uint32_t)acc * P::UNP) & C::MASK;
        umad_c(acc, m[k], P::UQ[0]);
        acc >>= W;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) {
            umad(acc, a[i], b[k - i]);
            if constexpr (TWO) umad(acc, c[i], d[k - i]);
        }
#pragma unroll
        for (int i = k - NL + 1; i < NL; ++i) umad_c(acc, m[i], P::UQ[k - i]);
        r[k - NL] = (uint32_t)acc & C::MASK;
        acc >>= W;
    }
    r[NL - 1] = (uint32_t)acc;
}
// square: off-diagonal products taken once against the doubled operand
template <class P>
MSM_HD void umont_sqr(uint32_t* __restrict__ r, const uint32_t* a) {
    using C = UC<P>;
    constexpr int NL = C::NL;
    constexpr int W = C::W;
    uint32_t m[NL], d2[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) d2[i] = a[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * NL - 1; ++k) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            int j = k - i;
            if (j > i && j < NL) umad(acc, a[i], d2[j]);
        }
        if ((k & 1) == 0) umad(acc, a[k / 2], a[k / 2]);
        if (k < NL) {
#pragma unroll
            for (int i = 0; i < k; ++i) umad_c(acc, m[i], P::UQ[k - i]);
            m[k] = ((uint32_t)acc * P::UNP) & C::MASK;
            umad_c(acc, m[k], P::UQ[0]);
        } else {
#pragma unroll
            for (int i = k - NL + 1; i < NL; ++i) umad_c(acc, m[i], P::UQ[k - i]);
            r[k - NL] = (uint32_t)acc & C::MASK;
        }
        acc >>= W;
    }
    r[NL - 1] = (uint32_t)acc;
}

// The multiplier bodies do not depend on the operands' bounds (those only decide, at compile time, whether the
// columns fit), so ONE out-of-line copy per field can serve every call site.  The G1 kernels expand the
// products inline (no call overhead, registers stay live across products); the G2 translation unit defines
// MSM_U_OUTLINE_PRODUCTS: its formulas would otherwise inline ~35 base-field products per addition, and
// device functions beyond the 128 KB reach of a direct branch have hung this GPU in practice.
// Operands travel as scalar arguments so that they are passed in VGPRs (the first 32; the rest on the stack).
#if defined(MSM_U_OUTLINE_PRODUCTS) && defined(__HIP_DEVICE_COMPILE__)
template <class P>
struct UOut {
    using C = UC<P>;
    static constexpr int NL = C::NL;
    template <size_t>
    using limb_t = uint32_t;
    template <size_t... I>
    __device__ __noinline__ static typename C::Limbs mul_(std::index_sequence<I...>, limb_t<I>... a, limb_t<I>... b) {
        uint32_t x[NL] = {a...}, y[NL] = {b...};
        typename C::Limbs r;
        umont<P, false>(r.v, x, y, nullptr, nullptr);
        return r;
    }
    template <size_t... I>
    __device__ __noinline__ static typename C::Limbs mul2_(std::index_sequence<I...>, limb_t<I>... a, limb_t<I>... b, limb_t<I>... c, limb_t<I>... d) {
        uint32_t x[NL] = {a...}, y[NL] = {b...}, z[NL] = {c...}, w[NL] = {d...};
        typename C::Limbs r;
        umont<P, true>(r.v, x, y, z, w);
        return r;
    }
    template <size_t... I>
    __device__ __noinline__ static typename C::Limbs sqr_(std::index_sequence<I...>, limb_t<I>... a) {
        uint32_t x[NL] = {a...};
        typename C::Limbs r;
        umont_sqr<P>(r.v, x);
        return r;
    }
    template <size_t... I>
    MSM_D static void mul(std::index_sequence<I...> q, uint32_t* r, const uint32_t* a, const uint32_t* b) {
        typename C::Limbs t = mul_(q, a[I]..., b[I]...);
        ((r[I] = t.v[I]), ...);
    }
    template <size_t... I>
    MSM_D static void mul2(std::index_sequence<I...> q, uint32_t* r, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d) {
        typename C::Limbs t = mul2_(q, a[I]..., b[I]..., c[I]..., d[I]...);
        ((r[I] = t.v[I]), ...);
    }
    template <size_t... I>
    MSM_D static void sqr(std::index_sequence<I...> q, uint32_t* r, const uint32_t* a) {
        typename C::Limbs t = sqr_(q, a[I]...);
        ((r[I] = t.v[I]), ...);
    }
};
#define MSM_UMONT(P, r, a, b) UOut<P>::mul(std::make_index_sequence<UC<P>::NL>{}, r, a, b)
#define MSM_UMONT2(P, r, a, b, c, d) UOut<P>::mul2(std::make_index_sequence<UC<P>::NL>{}, r, a, b, c, d)
#define MSM_UMONT_SQR(P, r, a) UOut<P>::sqr(std::make_index_sequence<UC<P>::NL>{}, r, a)
#else
#define MSM_UMONT(P, r, a, b) umont<P, false>(r, a, b, nullptr, nullptr)
#define MSM_UMONT2(P, r, a, b, c, d) umont<P, true>(r, a, b, c, d)
#define MSM_UMONT_SQR(P, r, a) umont_sqr<P>(r, a)
#endif

template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB>
MSM_HD auto fmul(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b) {
    using C = UC<P>;
    if constexpr (!C::cols_fit(LA, LB, 0, 0)) {
        // normalise the wider operand (a parallel carry pass) and try again
        if constexpr (LA >= LB) {
            static_assert(C::MASK + (LA >> C::W) < LA, "operands cannot be normalised further");
            return fmul(fnorm(a), b);
        } else {
            return fmul(a, fnorm(b));
        }
    } else {
        FuM<P, C::mul_vb((uint64_t)VA * VB)> r;
        MSM_UMONT(P, r.v, a.v, b.v);
        return r;
    }
}
template <class P, uint32_t LA, uint32_t VA>
MSM_HD auto fsqr(const Fu<P, LA, VA>& a) {
    using C = UC<P>;
    if constexpr (!C::cols_fit(LA, LA, 0, 0) || 2ull * LA >= (1ull << 32)) {
        static_assert(C::MASK + (LA >> C::W) < LA, "operand cannot be normalised further");
        return fsqr(fnorm(a));
    } else {
        FuM<P, C::mul_vb((uint64_t)VA * VA)> r;
        MSM_UMONT_SQR(P, r.v, a.v);
        return r;
    }
}
// (a*b + c*d) / R' with ONE reduction
template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB, uint32_t LC, uint32_t VC, uint32_t LD, uint32_t VD>
MSM_HD auto fmul2(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b, const Fu<P, LC, VC>& c, const Fu<P, LD, VD>& d) {
    using C = UC<P>;
    if constexpr (!C::cols_fit(LA, LB, LC, LD)) {
        constexpr uint64_t ab = (uint64_t)LA * LB, cd = (uint64_t)LC * LD;
        if constexpr (ab >= cd) {
            if constexpr (LA >= LB) {
                static_assert(C::MASK + (LA >> C::W) < LA, "operands cannot be normalised further");
                return fmul2(fnorm(a), b, c, d);
            } else {
                static_assert(C::MASK + (LB >> C::W) < LB, "operands cannot be normalised further");
                return fmul2(a, fnorm(b), c, d);
            }
        } else {
            if constexpr (LC >= LD) {
                static_assert(C::MASK + (LC >> C::W) < LC, "operands cannot be normalised further");
                return fmul2(a, b, fnorm(c), d);
            } else {
                static_assert(C::MASK + (LD >> C::W) < LD, "operands cannot be normalised further");
                return fmul2(a, b, c, fnorm(d));
            }
        }
    } else {
        FuM<P, C::mul_vb((uint64_t)VA * VB + (uint64_t)VC * VD)> r;
        MSM_UMONT2(P, r.v, a.v, b.v, c.v, d.v);
        return r;
    }
}

// zero test for strictly normalised elements (Montgomery outputs): the value is one of 0, q, .., (VB-1) q
template <class P, uint32_t J>
MSM_HD bool u_equals_jq(const uint32_t* a) {
    using C = UC<P>;
    constexpr typename C::Limbs qj = C::mulq(J);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < C::NL; ++i) o |= a[i] ^ qj.v[i];
    return o == 0;
}
// j*q for distinct j < 2^W differ in their lowest limb (q is odd): one compare per candidate, the full
// comparison only on a hit
template <class P, uint32_t... J>
MSM_HD bool u_is_multiple_of_q(const uint32_t* a, std::integer_sequence<uint32_t, J...>) {
    using C = UC<P>;
    const uint32_t a0 = a[0];
    const bool hit = ((a0 == C::mulq(J).v[0]) || ...);
    if (!hit) return false;
    return (u_equals_jq<P, J>(a) || ...);
}
template <class P, uint32_t VA>
MSM_HD bool fis_zero(const Fu<P, UC<P>::MASK, VA>& a) {
    static_assert(VA <= 64, "zero test wants a small value bound");
    return u_is_multiple_of_q<P>(a.v, std::make_integer_sequence<uint32_t, VA>{});
}

// ------------------------------------------------------------------------------------------
// boundary with the packed representation (Fp<P>: N 32-bit limbs, radix R = 2^(32N))
// ------------------------------------------------------------------------------------------
// A canonical packed x (< q, which the reference's own arithmetic presupposes of its inputs): the same bits shifted
// left by USHIFT are x*R'/R, a strictly normalised lazy element of value < 2^USHIFT * q.
template <class P>
using FuIn = Fu<P, UC<P>::MASK, (1u << P::USHIFT)>;

template <class P>
MSM_HD FuIn<P> u_from_packed(const uint32_t* x) {
    using C = UC<P>;
    constexpr int NL = C::NL, W = C::W, N = P::N, SH = P::USHIFT;
    FuIn<P> r;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int bit = W * i - SH;   // first bit of x in this limb
        uint32_t t;
        if (bit < 0) {
            t = x[0] << (-bit);
        } else {
            const int w = bit >> 5, o = bit & 31;
            uint32_t lo = w < N ? x[w] : 0u;
            uint32_t hi = (w + 1) < N ? x[w + 1] : 0u;
            t = o == 0 ? lo : ((lo >> o) | (hi << (32 - o)));
        }
        r.v[i] = (i < NL - 1) ? (t & C::MASK) : t;
    }
    return r;
}
// -> canonical packed image (value in [0, q), radix R)
template <class P, uint32_t LA, uint32_t VA>
MSM_HD void u_to_packed(uint32_t* out, const Fu<P, LA, VA>& a) {
    using C = UC<P>;
    constexpr int NL = C::NL, W = C::W, N = P::N;
    Fu<P, C::MASK, 1> k;
#pragma unroll
    for (int i = 0; i < NL; ++i) k.v[i] = P::UTO32[i];
    auto y = fmul(a, k);   // x*R mod q, strictly normalised
    constexpr int REPS = (int)decltype(y)::VB - 1;
    // subtract q while >= q
#pragma unroll
    for (int rep = 0; rep < REPS; ++rep) {
        uint32_t t[NL];
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            uint32_t s = y.v[i] - P::UQ[i] - borrow;
            borrow = s >> 31;
            t[i] = (i < NL - 1) ? (s & C::MASK) : s;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) y.v[i] = borrow ? y.v[i] : t[i];
    }
    // repack W-bit limbs into 32-bit words
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int bit = 32 * j;
        const int l = bit / W, o = bit % W;
        uint64_t v = (uint64_t)y.v[l] >> o;
        int have = W - o;
        if (l + 1 < NL) v |= (uint64_t)y.v[l + 1] << have;
        have += W;
        if (have < 32 && l + 2 < NL) v |= (uint64_t)y.v[l + 2] << have;
        out[j] = (uint32_t)v;
    }
}

// ------------------------------------------------------------------------------------------
// quadratic extension Fp[u]/(u^2+1) over Fu; both components share one type
// ------------------------------------------------------------------------------------------
template <class A>
struct Fp2u {
    using Base = A;
    using Params = typename A::Params;
    A c0, c1;
};
template <class P, uint32_t L0, uint32_t V0, uint32_t L1, uint32_t V1>
MSM_HD auto fp2u_make(const Fu<P, L0, V0>& a, const Fu<P, L1, V1>& b) {
    using T = Fu<P, (L0 > L1 ? L0 : L1), (V0 > V1 ? V0 : V1)>;
    return Fp2u<T>{uwiden<T>(a), uwiden<T>(b)};
}
template <class T, class A>
MSM_HD Fp2u<T> uwiden2(const Fp2u<A>& a) {
    return {uwiden<T>(a.c0), uwiden<T>(a.c1)};
}
template <class A, class B>
MSM_HD auto fadd(const Fp2u<A>& a, const Fp2u<B>& b) { return fp2u_make(fadd(a.c0, b.c0), fadd(a.c1, b.c1)); }
template <class A>
MSM_HD auto fdbl(const Fp2u<A>& a) { return fp2u_make(fdbl(a.c0), fdbl(a.c1)); }
template <class A, class B>
MSM_HD auto fsub(const Fp2u<A>& a, const Fp2u<B>& b) { return fp2u_make(fsub(a.c0, b.c0), fsub(a.c1, b.c1)); }
template <class A>
MSM_HD auto fneg(const Fp2u<A>& a) { return fp2u_make(fneg(a.c0), fneg(a.c1)); }
template <class A>
MSM_HD auto fcneg(const Fp2u<A>& a, bool f) { return fp2u_make(fcneg(a.c0, f), fcneg(a.c1, f)); }
template <class A>
MSM_HD auto fnorm(const Fp2u<A>& a) { return fp2u_make(fnorm(a.c0), fnorm(a.c1)); }
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u : four half products, two reductions
template <class A, class B>
MSM_HD auto fmul(const Fp2u<A>& a, const Fp2u<B>& b) {
    auto r0 = fmul2(a.c0, b.c0, a.c1, fneg(b.c1));
    auto r1 = fmul2(a.c0, b.c1, a.c1, b.c0);
    return fp2u_make(r0, r1);
}
// (a0 + a1)(a0 - a1) + 2 a0 a1 u.  Where R'/q is small (BN254: 169) the value bound of the first product,
// (2V)(2V+1), would not let the accumulation loop close; there the real part is taken as a0*a0 + a1*(-a1)
// with one reduction (bound 2V^2), at the price of one more half product.
template <class A>
MSM_HD auto fsqr(const Fp2u<A>& a) {
    using C = UC<typename A::Params>;
    auto r1 = fmul(fdbl(a.c0), a.c1);
    if constexpr (C::RQ >= 1024) {
        auto r0 = fmul(fadd(a.c0, a.c1), fsub(a.c0, a.c1));
        return fp2u_make(r0, r1);
    } else {
        auto r0 = fmul2(a.c0, a.c0, a.c1, fneg(a.c1));
        return fp2u_make(r0, r1);
    }
}
// (a*b + c*d) over Fp2 is not used; the fused form of the curve code falls back to two products
template <class A>
MSM_HD bool fis_zero(const Fp2u<A>& a) { return fis_zero(a.c0) && fis_zero(a.c1); }


// ------------------------------------------------------------------------------------------
// Fp2 element DISTRIBUTED over a pair of adjacent lanes: the even lane holds c0, the odd lane c1.
// Why: one lane per Fp2 point needs > 256 registers (one wave per SIMD, products out of line); split over two lanes
// every value is half as wide, the code is fully inline at ~3 waves per SIMD, and a bucket is served by two lanes.
// Component-wise operations need no communication; a product exchanges the partner's components with one DPP move
// per limb (quad_perm [1,0,3,2]) and each lane then does ONE double-width accumulation:
//   even lane:  c0 = a0*b0 + a1*(-b1)        odd lane:  c1 = a1*b0 + a0*b1      i.e.  own_a*U + partner_a*V.
// Both lanes of a pair always execute the same path (every predicate below is reduced over the pair).
// ------------------------------------------------------------------------------------------
MSM_D uint32_t pair_parity() {
#if defined(__HIP_DEVICE_COMPILE__)
    return threadIdx.x & 1u;
#else
    return 0;
#endif
}
MSM_D uint32_t pair_xchg(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
#else
    return x;
#endif
}
template <class P, uint32_t L, uint32_t V>
MSM_D Fu<P, L, V> pair_xchg(const Fu<P, L, V>& a) {
    Fu<P, L, V> r;
#pragma unroll
    for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = pair_xchg(a.v[i]);
    return r;
}
// odd ? b : a   (both re-typed to the looser of the two bounds)
template <class P, uint32_t LA, uint32_t VA, uint32_t LB, uint32_t VB>
MSM_D auto pair_select(const Fu<P, LA, VA>& a, const Fu<P, LB, VB>& b, bool odd) {
    Fu<P, (LA > LB ? LA : LB), (VA > VB ? VA : VB)> r;
#pragma unroll
    for (int i = 0; i < UC<P>::NL; ++i) r.v[i] = odd ? b.v[i] : a.v[i];
    return r;
}
template <class A>
struct Fp2d {
    using Base = A;
    using Params = typename A::Params;
    A v;   // this lane's component
};
template <class A, class B>
MSM_D auto fadd(const Fp2d<A>& a, const Fp2d<B>& b) { auto r = fadd(a.v, b.v); return Fp2d<decltype(r)>{r}; }
template <class A>
MSM_D auto fdbl(const Fp2d<A>& a) { auto r = fdbl(a.v); return Fp2d<decltype(r)>{r}; }
template <class A, class B>
MSM_D auto fsub(const Fp2d<A>& a, const Fp2d<B>& b) { auto r = fsub(a.v, b.v); return Fp2d<decltype(r)>{r}; }
template <class A>
MSM_D auto fneg(const Fp2d<A>& a) { auto r = fneg(a.v); return Fp2d<decltype(r)>{r}; }
template <class A>
MSM_D auto fcneg(const Fp2d<A>& a, bool f) { auto r = fcneg(a.v, f); return Fp2d<decltype(r)>{r}; }
template <class A>
MSM_D auto fnorm(const Fp2d<A>& a) { auto r = fnorm(a.v); return Fp2d<decltype(r)>{r}; }
template <class A, class B>
MSM_D auto fmul(const Fp2d<A>& a, const Fp2d<B>& b) {
    const bool odd = pair_parity() != 0;
    const A pa = pair_xchg(a.v);
    const B pb = pair_xchg(b.v);
    auto U = pair_select(b.v, pb, odd);            // even: own b (b0), odd: partner b (b0)
    auto Vv = pair_select(fneg(pb), b.v, odd);     // even: -b1,         odd: own b (b1)
    auto r = fmul2(a.v, U, pa, Vv);
    return Fp2d<decltype(r)>{r};
}
template <class A>
MSM_D auto fsqr(const Fp2d<A>& a) {
    using C = UC<typename A::Params>;
    if constexpr (C::RQ >= 1024) {
        // even: (a0 + a1)(a0 - a1),  odd: (2 a0) a1   -- one single-width product per lane
        const bool odd = pair_parity() != 0;
        const A pa = pair_xchg(a.v);
        auto X = pair_select(fadd(a.v, pa), fdbl(pa), odd);
        auto Y = pair_select(fsub(a.v, pa), a.v, odd);
        auto r = fmul(X, Y);
        return Fp2d<decltype(r)>{r};
    } else {
        return fmul(a, a);   // tight radix (BN254): the bound of (2V)(2V+1) would not close, see Fp2u::fsqr
    }
}
template <class A>
MSM_D bool fis_zero(const Fp2d<A>& a) {
    const uint32_t z = fis_zero(a.v) ? 1u : 0u;
    return (z & pair_xchg(z)) != 0;
}

#endif  // __HIPCC__

}  // namespace msm
