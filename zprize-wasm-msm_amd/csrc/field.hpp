// field.hpp -- Montgomery prime-field and quadratic-extension arithmetic for gfx950 (and host).
//
// Replaces, on the device, the reference's WASM field layer:
//   wasmcurves/src/build_f1m.js  (f1m_add :71-89, f1m_sub :91-105, f1m_neg :107-117,
//                                 f1m_mul :466-777, f1m_square :779-1076, isOne :1318-1327)
//   wasmcurves/src/build_f2m.js  (f2m_mul :152-188, f2m_square :211-252, add/sub/neg :34-112)
//   wasmcurves/src/build_int.js  (int_add :186-228, int_sub :232-277, int_gte :148-181)
// Same value semantics: elements are n32 little-endian 32-bit limbs in Montgomery form with
// R = 2^(32*N), every result fully reduced to [0, q).  The byte image of an element is therefore
// identical to the reference's linear-memory image.
//
// On the DEVICE values are kept "relaxed" in [0, 2q) between operations (see tools/gen_fp_asm.py): the
// Montgomery product needs no final subtraction because R > 4q for both moduli, add/sub correct by 2q,
// and canon() restores the reference's [0, q) representation wherever a value leaves the engine.
// The HOST build of the same templates stays canonical throughout.
//
// This packed field serves the engine's auxiliary kernels (conversions, checks, generated points, FFTs) and, in its
// 64-bit twin field_host.hpp, the host combine; the ALU-bound bucket kernels use the unsaturated form of field_u.hpp.
// Device multiplier: product-scanning Montgomery on 32-bit limbs.  v_mad_u64_u32 issues at full
// rate on gfx950 (profiles/r01_valu_rates_microbench.txt), so one instruction does a 32x32
// multiply and a 64-bit accumulate; its carry-out is folded into a third accumulator word with
// one v_addc_co_u32.  The host build of the same templates uses portable 64-bit C.
#pragma once
#include <cstdint>
#include <utility>
#include <hip/hip_runtime.h>

#include "params_gen.hpp"
#include "fp_asm_gen.hpp"

#define MSM_HD __host__ __device__ __forceinline__
#define MSM_D __device__ __forceinline__
// Largest total of by-value struct operands an out-of-line (__noinline__) device function may take: two 384-byte XYZZ<Fp2> structs
// passed by value arrived corrupted in the callee on gfx950 / ROCm 7.2, 256-byte ones were fine.  Static
// asserts at the by-value functions and tools/check_code_objects.py (source scan) enforce it.
#define MSM_BYVALUE_MAX 256

namespace msm {

// (hi:lo) += a*b, 96-bit accumulator
MSM_HD void mac(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "v"(b)
        : "vcc");
#else
    uint64_t p = (uint64_t)a * b;
    lo += p;
    hi += (lo < p);
#endif
}
// same with b a wave-uniform constant kept in an SGPR
MSM_HD void mac_s(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi)
        : "v"(a), "s"(b)
        : "vcc");
#else
    mac(lo, hi, a, b);
#endif
}

// INL = true: multiplies are expanded inline at every use; INL = false: one out-of-line body per translation unit
// (small code) -- what the kernels of conv_kernels.hip and fft_kernels.hip use.  The bucket kernels no longer compute in this packed
// representation at all (field_u.hpp); Fp<P, true> survives as the tag under which their G1 instantiations are named.
// Both flavours have the same layout and value semantics.
template <class P, bool INL = false>
struct Fp {
    static constexpr int N = P::N;
    static constexpr int BYTES = 4 * P::N;
    using Params = P;
    uint32_t v[N];

    MSM_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    MSM_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = P::ONE[i];
        return r;
    }
    // all limbs zero (the only representation of 0 on the host; one of two on the device)
    MSM_HD bool is_zero_exact() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) o |= v[i];
        return o == 0;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // relaxed: 0 is represented by 0 or by q
    MSM_D bool is_zero() const {
        uint32_t o = 0, oq = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            o |= v[i];
            oq |= v[i] ^ P::Q[i];
        }
        return o == 0 || oq == 0;
    }
    MSM_D static bool eq(const Fp& a, const Fp& b) { return sub(a, b).is_zero(); }
    MSM_D bool is_one() const { return eq(*this, one()); }
#else
    bool is_zero() const { return is_zero_exact(); }
    bool is_one() const {
        uint32_t o = 0;
        for (int i = 0; i < N; ++i) o |= v[i] ^ P::ONE[i];
        return o == 0;
    }
    static bool eq(const Fp& a, const Fp& b) {
        uint32_t o = 0;
        for (int i = 0; i < N; ++i) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }
#endif

    // r = a - q if a >= q else a   (a < 2q, carry = bit 32N of a)
    MSM_HD static Fp reduce_once(const Fp& a, uint32_t carry) {
        Fp d;
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] - P::Q[i] - br;
            d.v[i] = (uint32_t)t;
            br = (t >> 32) & 1;
        }
        bool use_d = (carry != 0) || (br == 0);
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = use_d ? d.v[i] : a.v[i];
        return r;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // r_i = (lane in mask) ? d_i : r_i
    MSM_D static Fp select(Fp r, const Fp& d, uint64_t mask) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r.v[i]) : "v"(d.v[i]), "s"(mask));
        return r;
    }
    // relaxed [0, 2q) -> canonical [0, q)
    MSM_D static Fp canon(const Fp& a) {
        Fp d = a;
        uint64_t ge = FpAsm<P>::add_minus_q(d.v);
        return select(a, d, ge);
    }
#else
    static Fp canon(const Fp& a) { return a; }
#endif

    // build_f1m.js:71-89
    MSM_HD static Fp add(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r;
        (void)FpAsm<P>::add_raw(r.v, a.v, b.v);          // < 4q < 2^(32N): no carry out
        Fp d = r;
        uint64_t ge = FpAsm<P>::add_minus_2q(d.v);       // d = r - 2q where r >= 2q
        return select(r, d, ge);
#else
        Fp s;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] + b.v[i] + c;
            s.v[i] = (uint32_t)t;
            c = t >> 32;
        }
        return reduce_once(s, (uint32_t)c);
#endif
    }
    MSM_HD static Fp dbl(const Fp& a) { return add(a, a); }
    // build_f1m.js:91-105
    MSM_HD static Fp sub(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r;
        uint64_t borrow = FpAsm<P>::sub_raw(r.v, a.v, b.v);
        (void)FpAsm<P>::add_2q_masked(r.v, borrow);      // (-2q, 2q) -> [0, 2q)
        return r;
#else
        Fp d;
        uint64_t br = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)a.v[i] - b.v[i] - br;
            d.v[i] = (uint32_t)t;
            br = (t >> 32) & 1;
        }
        uint32_t mask = (uint32_t)0 - (uint32_t)br;
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            uint64_t t = (uint64_t)d.v[i] + (P::Q[i] & mask) + c;
            d.v[i] = (uint32_t)t;
            c = t >> 32;
        }
        return d;
#endif
    }
    // build_f1m.js:107-117 (0 - x; neg(0) == 0)
    MSM_HD static Fp neg(const Fp& a) { return cneg(a, true); }
    MSM_HD static Fp cneg(const Fp& a, bool flag) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r = a;
        FpAsm<P>::rsub_2q_masked(r.v, __ballot(flag && !a.is_zero_exact()));   // 2q - a on the flagged lanes; -0 = 0
        return r;
#else
        Fp n = sub(zero(), a);
        Fp r;
#pragma unroll
        for (int i = 0; i < N; ++i) r.v[i] = flag ? n.v[i] : a.v[i];
        return r;
#endif
    }

    // build_f1m.js:466-777 -- Montgomery product a*b*R^-1 mod q, fully reduced.
    MSM_HD static Fp mul_inl(const Fp& a, const Fp& b) {
        uint32_t m[N];
        Fp r;
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) mac(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
            for (int i = 0; i < k; ++i) mac_s(lo, hi, m[i], P::Q[k - i]);
            m[k] = (uint32_t)lo * P::NP32;
            mac_s(lo, hi, m[k], P::Q[0]);
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#pragma unroll
        for (int k = N; k < 2 * N; ++k) {
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) mac(lo, hi, a.v[i], b.v[k - i]);
#pragma unroll
            for (int i = k - N + 1; i < N; ++i) mac_s(lo, hi, m[i], P::Q[k - i]);
            r.v[k - N] = (uint32_t)lo;
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        return r;   // relaxed inputs (< 2q) give t < 2q < 2^(32N); canonicalised only at the engine's outputs
#else
        return reduce_once(r, (uint32_t)lo);
#endif
    }
    // build_f1m.js:779-1076 -- square: off-diagonal products computed once and doubled.
    MSM_HD static Fp sqr_inl(const Fp& a) {
        uint32_t m[N];
        Fp r;
        uint64_t lo = 0;   // running (reduction + carry) accumulator
        uint32_t hi = 0;
#pragma unroll
        for (int k = 0; k < 2 * N; ++k) {
            // off-diagonal sum for this column: sum_{i<j, i+j=k} a_i a_j
            uint64_t slo = 0;
            uint32_t shi = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                int j = k - i;
                if (j > i && j < N) mac(slo, shi, a.v[i], a.v[j]);
            }
            // double it and add to the running accumulator
            uint32_t s0 = (uint32_t)slo, s1 = (uint32_t)(slo >> 32);
            uint32_t d0 = s0 << 1, d1 = (s1 << 1) | (s0 >> 31), d2 = (shi << 1) | (s1 >> 31);
            uint64_t t = (uint64_t)(uint32_t)lo + d0;
            uint32_t l0 = (uint32_t)t;
            t = (t >> 32) + (uint32_t)(lo >> 32) + d1;
            uint32_t l1 = (uint32_t)t;
            hi = hi + d2 + (uint32_t)(t >> 32);
            lo = ((uint64_t)l1 << 32) | l0;
            if ((k & 1) == 0) mac(lo, hi, a.v[k / 2], a.v[k / 2]);
            if (k < N) {
#pragma unroll
                for (int i = 0; i < k; ++i) mac_s(lo, hi, m[i], P::Q[k - i]);
                m[k] = (uint32_t)lo * P::NP32;
                mac_s(lo, hi, m[k], P::Q[0]);
            } else {
#pragma unroll
                for (int i = k - N + 1; i < N; ++i) mac_s(lo, hi, m[i], P::Q[k - i]);
                r.v[k - N] = (uint32_t)lo;
            }
            lo = (lo >> 32) | ((uint64_t)hi << 32);
            hi = 0;
        }
        #if defined(__HIP_DEVICE_COMPILE__)
        return r;
#else
        return reduce_once(r, (uint32_t)lo);
#endif
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // Out-of-line bodies take their limbs as 2N (or N) scalar arguments so that both operands travel in
    // VGPRs (a by-value struct argument beyond the first would be passed through scratch memory).
    template <size_t>
    using limb_t = uint32_t;
    template <size_t... I>
    __device__ __noinline__ static Fp mul_ni_(std::index_sequence<I...>, limb_t<I>... a, limb_t<I>... b) {
        Fp x{{a...}}, y{{b...}};
        return mul_inl(x, y);
    }
    template <size_t... I>
    __device__ __noinline__ static Fp sqr_ni_(std::index_sequence<I...>, limb_t<I>... a) {
        Fp x{{a...}};
        return sqr_inl(x);
    }
    template <size_t... I>
    MSM_D static Fp mul_ni_x(std::index_sequence<I...> q, const Fp& a, const Fp& b) { return mul_ni_(q, a.v[I]..., b.v[I]...); }
    template <size_t... I>
    MSM_D static Fp sqr_ni_x(std::index_sequence<I...> q, const Fp& a) { return sqr_ni_(q, a.v[I]...); }
    MSM_D static Fp mul_ni(const Fp& a, const Fp& b) { return mul_ni_x(std::make_index_sequence<N>{}, a, b); }
    MSM_D static Fp sqr_ni(const Fp& a) { return sqr_ni_x(std::make_index_sequence<N>{}, a); }
    MSM_D static Fp mul(const Fp& a, const Fp& b) {
        if constexpr (INL) return mul_inl(a, b);
        else return mul_ni(a, b);
    }
    MSM_D static Fp sqr(const Fp& a) {
        if constexpr (INL) return sqr_inl(a);
        else return sqr_ni(a);
    }
#else
    static Fp mul(const Fp& a, const Fp& b) { return mul_inl(a, b); }
    static Fp sqr(const Fp& a) { return sqr_inl(a); }
#endif

    // x^(q-2); used only off the hot path (normalisation, batched affine conversion)
    MSM_HD static Fp inv(const Fp& x) {
        Fp acc = one();
        Fp base = x;
        for (int i = 0; i < 32 * N; ++i) {
            if ((P::QM2[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    // build_f1m.js:1089-1110
    MSM_HD static Fp to_mont(const Fp& x) {
        Fp r2;
#pragma unroll
        for (int i = 0; i < N; ++i) r2.v[i] = P::R2[i];
        return mul(x, r2);
    }
    MSM_HD static Fp from_mont(const Fp& x) {
        Fp o = zero();
        o.v[0] = 1;
        return mul(x, o);
    }
};

// Fp2 = Fp[u]/(u^2 + 1): both reference curves pass mulNonResidueFn = "f1m_neg"
// (bls12381/build_bls12381.js:48, bn128/build_bn128.js:43).
template <class B_>
struct Fp2T {
    using B = B_;
    using Params = typename B::Params;
    static constexpr int BYTES = 2 * B::BYTES;
    B c0, c1;

    MSM_HD static Fp2T zero() { return {B::zero(), B::zero()}; }
    MSM_HD static Fp2T one() { return {B::one(), B::zero()}; }
    MSM_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    MSM_HD bool is_one() const { return c0.is_one() && c1.is_zero(); }   // build_f2m.js:432-449
    MSM_HD static bool eq(const Fp2T& a, const Fp2T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
    MSM_HD static Fp2T add(const Fp2T& a, const Fp2T& b) { return {B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    MSM_HD static Fp2T dbl(const Fp2T& a) { return add(a, a); }
    MSM_HD static Fp2T sub(const Fp2T& a, const Fp2T& b) { return {B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    MSM_HD static Fp2T neg(const Fp2T& a) { return {B::neg(a.c0), B::neg(a.c1)}; }
    MSM_HD static Fp2T canon(const Fp2T& a) { return {B::canon(a.c0), B::canon(a.c1)}; }
    MSM_HD static Fp2T cneg(const Fp2T& a, bool f) { return {B::cneg(a.c0, f), B::cneg(a.c1, f)}; }
    // build_f2m.js:152-188 (Karatsuba, 3 base multiplications)
    MSM_HD static Fp2T mul(const Fp2T& a, const Fp2T& b) {
        B A = B::mul(a.c0, b.c0);
        B Bb = B::mul(a.c1, b.c1);
        B C = B::mul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        return {B::sub(A, Bb), B::sub(C, B::add(A, Bb))};
    }
    // build_f2m.js:211-252: (c0+c1)(c0-c1), 2 c0 c1
    MSM_HD static Fp2T sqr(const Fp2T& a) {
        B ab = B::mul(a.c0, a.c1);
        B r0 = B::mul(B::add(a.c0, a.c1), B::sub(a.c0, a.c1));
        return {r0, B::add(ab, ab)};
    }
    // build_f2m.js:378-408
    MSM_HD static Fp2T inv(const Fp2T& a) {
        B t = B::inv(B::add(B::sqr(a.c0), B::sqr(a.c1)));
        return {B::mul(a.c0, t), B::neg(B::mul(a.c1, t))};
    }
};

template <class P>
using Fp2 = Fp2T<Fp<P>>;

}  // namespace msm
