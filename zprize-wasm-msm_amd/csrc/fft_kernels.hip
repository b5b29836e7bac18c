// fft_kernels.hip -- batchApplyKey and the FFTs over Fr, G1 and G2 (build_applykey.js, build_fft.js).
#include "engine_ctx.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------
// batchApplyKey (build_applykey.js:20-76 wired at bls12381/build_bls12381.js:79-92, bn128/build_bn128.js:84-88):
//   out[i] = (first * inc^i) * in[i],  first / inc in Fr, Montgomery form.
// The reference walks i sequentially (t = t * inc); here lane i raises inc to the i-th power itself (<= 2 * 27 Montgomery
// products in Fr), leaves Montgomery form and runs a 255-step double-and-add on its own point.
// Fr arithmetic: plain CIOS Montgomery on 8 x 32-bit limbs, always fully reduced (as frm_mul, build_f1m.js:466-777 over r).
// ------------------------------------------------------------------------------------------
template <class P>
MSM_D void fr_mul(const uint32_t* a, const uint32_t* b, uint32_t* r) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) t[i] = 0;
    for (int i = 0; i < 8; ++i) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c += (uint64_t)a[j] * b[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * P::FR_NP32;
        c = (uint64_t)m * P::RORDER[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            c += (uint64_t)m * P::RORDER[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    // t < 2r: one conditional subtraction
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t v = (uint64_t)t[i] - P::RORDER[i] - br;
        d[i] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    const bool use_d = t[8] != 0 || br == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = use_d ? d[i] : t[i];
}
// s = first * inc^i (Montgomery)
template <class P>
MSM_D void fr_key(const uint32_t* first, const uint32_t* inc, uint32_t i, uint32_t* s) {
    uint32_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s[k] = first[k];
        e[k] = inc[k];
    }
    while (i) {
        if (i & 1u) fr_mul<P>(s, e, s);
        fr_mul<P>(e, e, e);
        i >>= 1;
    }
}
template <class P>
__global__ void __launch_bounds__(256) k_fr_batch_apply_key(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ key, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s[8], x[8];
    fr_key<P>(key, key + 8, i, s);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = in[(size_t)i * 8 + k];
    fr_mul<P>(x, s, x);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(size_t)i * 8 + k] = x[k];
}
template <class F, bool JAC_IN>
__global__ void __launch_bounds__(256) k_batch_apply_key(const uint8_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ key, uint8_t* __restrict__ out) {
    using P = typename F::Params;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p;
    if constexpr (JAC_IN) {
        Jacobian<F> j = load16<Jacobian<F>>(in + (size_t)i * sizeof(Jacobian<F>));
        p = XYZZ<F>::to_affine(XYZZ<F>::from_jacobian(j));   // infinity -> (0, 0)
    } else {
        p = load16<Affine<F>>(in + (size_t)i * sizeof(Affine<F>));
    }
    uint32_t s[8], k[8];
    fr_key<P>(key, key + 8, i, s);
    uint32_t one[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    fr_mul<P>(s, one, k);   // frm_fromMontgomery (buildGTimesFr)
    XYZZ<F> acc = XYZZ<F>::inf();
    if (!p.is_inf()) {
        for (int bit = 255; bit >= 0; --bit) {
            acc = XYZZ<F>::dbl_ni(acc);
            if ((k[bit >> 5] >> (bit & 31)) & 1u) XYZZ<F>::madd(acc, p.x, p.y);
        }
    }
    Jacobian<F> o;
    if (acc.is_inf()) {
        o = {F::zero(), F::one(), F::zero()};
    } else {
        Affine<F> a = Affine<F>::canon(XYZZ<F>::to_affine(acc));
        o = {a.x, a.y, F::one()};
    }
    store16(out + (size_t)i * sizeof(Jacobian<F>), o);
}

// ------------------------------------------------------------------------------------------
// FFTs (build_fft.js:178-245 fft/ifft, :247-394 rawfft, :396-516 fftFinal, roots :33-50).
// Radix-2, decimation in time: bit-reversal permutation, then log2 n passes of n/2 butterflies
//   (u, t = w * x2) -> (u + t, u - t),  w = omega_n^(j * n/m)  from a table of the n/2 powers of omega_n;
// the inverse transform is the forward one followed by out[i] = x[(n - i) mod n] * 2^-bits.
// Over Fr the data are 32-byte Montgomery elements (byte-identical results); over G1/G2 the data are points (kept in
// XYZZ between the passes), "w * x2" is a 255-step double-and-add with the plain value of w (g?m_timesFr).
// ------------------------------------------------------------------------------------------
template <class P>
MSM_D void fr_add(const uint32_t* a, const uint32_t* b, uint32_t* r) {
    uint32_t s[8], d[8];
    uint64_t c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)a[i] + b[i];
        s[i] = (uint32_t)c;
        c >>= 32;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t v = (uint64_t)s[i] - P::RORDER[i] - br;
        d[i] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    const bool use_d = c != 0 || br == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = use_d ? d[i] : s[i];
}
template <class P>
MSM_D void fr_sub(const uint32_t* a, const uint32_t* b, uint32_t* r) {
    uint32_t d[8];
    uint64_t br = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint64_t v = (uint64_t)a[i] - b[i] - br;
        d[i] = (uint32_t)v;
        br = (v >> 32) & 1;
    }
    const uint32_t mask = (uint32_t)0 - (uint32_t)br;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c += (uint64_t)d[i] + (P::RORDER[i] & mask);
        r[i] = (uint32_t)c;
        c >>= 32;
    }
}
// tw[k] = omega_n^k for k < max(1, n/2); omega_n = FR_ROOT^(2^(MAXBITS - bits))
template <class P>
__global__ void __launch_bounds__(256) k_fft_twiddles(uint32_t bits, uint32_t count, uint32_t* __restrict__ tw) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    uint32_t w[8], s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w[i] = P::FR_ROOT[i];
        s[i] = P::FR_ONE[i];
    }
    for (uint32_t i = bits; i < (uint32_t)P::FR_MAXBITS; ++i) fr_mul<P>(w, w, w);
    fr_key<P>(s, w, k, s);
#pragma unroll
    for (int i = 0; i < 8; ++i) tw[(size_t)k * 8 + i] = s[i];
}
MSM_D uint32_t fft_rev(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }
template <class P>
__global__ void __launch_bounds__(256) k_fr_fft_bitrev(const uint32_t* __restrict__ in, uint32_t n, uint32_t bits, uint32_t* __restrict__ x) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* s = reinterpret_cast<const uint4*>(in) + (size_t)i * 2;
    uint4* d = reinterpret_cast<uint4*>(x) + (size_t)fft_rev(i, bits) * 2;
    d[0] = s[0];
    d[1] = s[1];
}
template <class P>
__global__ void __launch_bounds__(256) k_fr_fft_stage(uint32_t* __restrict__ x, const uint32_t* __restrict__ tw, uint32_t n, uint32_t bits, uint32_t s) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n / 2) return;
    const uint32_t h = 1u << (s - 1), j = idx & (h - 1), i1 = ((idx >> (s - 1)) << s) + j, i2 = i1 + h;
    uint32_t a[8], b[8], w[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = x[(size_t)i1 * 8 + k];
        b[k] = x[(size_t)i2 * 8 + k];
        w[k] = tw[(size_t)(j << (bits - s)) * 8 + k];
    }
    fr_mul<P>(b, w, t);
    fr_add<P>(a, t, b);
    fr_sub<P>(a, t, w);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        x[(size_t)i1 * 8 + k] = b[k];
        x[(size_t)i2 * 8 + k] = w[k];
    }
}
// 2^-bits in Montgomery form
template <class P>
MSM_D void fr_inv_pow2(uint32_t bits, uint32_t* r) {
    uint32_t h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        r[i] = P::FR_ONE[i];
        h[i] = P::FR_HALF[i];
    }
    for (uint32_t b = 0; b < bits; ++b) fr_mul<P>(r, h, r);
}
template <class P>
__global__ void __launch_bounds__(256) k_fr_fft_final(const uint32_t* __restrict__ x, uint32_t n, uint32_t bits, uint32_t inverse, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v[8];
    const size_t src = inverse ? ((n - i) & (n - 1)) : i;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = x[src * 8 + k];
    if (inverse) {
        uint32_t f[8];
        fr_inv_pow2<P>(bits, f);
        fr_mul<P>(v, f, v);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(size_t)i * 8 + k] = v[k];
}
// k * p for a point in XYZZ, k = plain value of the Montgomery element w
template <class F>
MSM_D XYZZ<F> g_times_fr(const XYZZ<F>& p, const uint32_t* w_mont) {
    using P = typename F::Params;
    uint32_t k[8], one[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    fr_mul<P>(w_mont, one, k);
    XYZZ<F> acc = XYZZ<F>::inf();
    if (p.is_inf()) return acc;
    for (int bit = 255; bit >= 0; --bit) {
        XYZZ<F> d;
        XYZZ<F>::dbl_to(d, acc);
        if ((k[bit >> 5] >> (bit & 31)) & 1u) XYZZ<F>::add_to(acc, d, p);
        else acc = d;
    }
    return acc;
}
template <class F>
__global__ void __launch_bounds__(256) k_g_fft_bitrev(const uint8_t* __restrict__ in, uint32_t n, uint32_t bits, uint8_t* __restrict__ x) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jacobian<F> j = load16<Jacobian<F>>(in + (size_t)i * sizeof(Jacobian<F>));
    store16(x + (size_t)fft_rev(i, bits) * sizeof(XYZZ<F>), XYZZ<F>::from_jacobian(j));
}
template <class F>
__global__ void __launch_bounds__(256) k_g_fft_stage(uint8_t* __restrict__ x, const uint32_t* __restrict__ tw, uint32_t n, uint32_t bits, uint32_t s) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n / 2) return;
    const uint32_t h = 1u << (s - 1), j = idx & (h - 1), i1 = ((idx >> (s - 1)) << s) + j, i2 = i1 + h;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = tw[(size_t)(j << (bits - s)) * 8 + k];
    XYZZ<F> u = load16<XYZZ<F>>(x + (size_t)i1 * sizeof(XYZZ<F>));
    XYZZ<F> t = g_times_fr<F>(load16<XYZZ<F>>(x + (size_t)i2 * sizeof(XYZZ<F>)), w);
    XYZZ<F> r1, r2;
    const XYZZ<F> nt = XYZZ<F>::neg(t);
    XYZZ<F>::add_to(r1, u, t);
    XYZZ<F>::add_to(r2, u, nt);
    store16(x + (size_t)i1 * sizeof(XYZZ<F>), r1);
    store16(x + (size_t)i2 * sizeof(XYZZ<F>), r2);
}
template <class F>
__global__ void __launch_bounds__(256) k_g_fft_final(const uint8_t* __restrict__ x, uint32_t n, uint32_t bits, uint32_t inverse, uint8_t* __restrict__ out) {
    using P = typename F::Params;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t src = inverse ? ((n - i) & (n - 1)) : i;
    XYZZ<F> p = load16<XYZZ<F>>(x + src * sizeof(XYZZ<F>));
    if (inverse) {
        uint32_t f[8];
        fr_inv_pow2<P>(bits, f);
        p = g_times_fr<F>(p, f);
    }
    Jacobian<F> o;
    if (p.is_inf()) {
        o = {F::zero(), F::one(), F::zero()};
    } else {
        Affine<F> a = Affine<F>::canon(XYZZ<F>::to_affine(p));
        o = {a.x, a.y, F::one()};
    }
    store16(out + (size_t)i * sizeof(Jacobian<F>), o);
}


// ------------------------------------------------------------------------------------------
// Building blocks of the blocked FFT (what ffjavascript drives for transforms larger than one WASM instance):
//   <p>_fftJoin / _fftJoinExt / _fftJoinExtInv / _prepareLagrangeEvaluation (build_fft.js:657-1112)  element-wise over two buffers,
//        with the running key W_i = first * inc^i (the reference walks i sequentially; lane i raises inc to the i-th power itself);
//   <p>_fftMix (:1114-1236) one radix-2 pass of span 2^exp;   <p>_fftFinal (:1238-1353) out[i] = factor * in[n-1-i].
// One element type per instantiation: FrEl (32-byte Montgomery scalars, byte-identical to the reference) or PtEl<F> (Jacobian points
// in, NORMALISED Jacobian points out; the scalar product is g?m_timesFr).  SM / SC = SHIFT_TO_M[totalBits] / SCONST[totalBits].
// ------------------------------------------------------------------------------------------
template <class P>
struct FrEl {
    using Params = P;
    static constexpr size_t BYTES = 32;
    uint32_t v[8];
    MSM_D static FrEl load(const uint8_t* p) {
        FrEl e;
        const uint4* s = reinterpret_cast<const uint4*>(p);
        uint4 a = s[0], b = s[1];
        e.v[0] = a.x; e.v[1] = a.y; e.v[2] = a.z; e.v[3] = a.w; e.v[4] = b.x; e.v[5] = b.y; e.v[6] = b.z; e.v[7] = b.w;
        return e;
    }
    MSM_D void store(uint8_t* p) const {
        uint4* d = reinterpret_cast<uint4*>(p);
        d[0] = make_uint4(v[0], v[1], v[2], v[3]);
        d[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
    MSM_D static FrEl add(const FrEl& a, const FrEl& b) { FrEl r; fr_add<P>(a.v, b.v, r.v); return r; }
    MSM_D static FrEl sub(const FrEl& a, const FrEl& b) { FrEl r; fr_sub<P>(a.v, b.v, r.v); return r; }
    MSM_D static FrEl times(const FrEl& a, const uint32_t* w_mont) { FrEl r; fr_mul<P>(a.v, w_mont, r.v); return r; }
};
template <class F>
struct PtEl {
    using Params = typename F::Params;
    static constexpr size_t BYTES = sizeof(Jacobian<F>);
    XYZZ<F> p;
    MSM_D static PtEl load(const uint8_t* q) { return {XYZZ<F>::from_jacobian(load16<Jacobian<F>>(q))}; }
    MSM_D void store(uint8_t* q) const {
        Jacobian<F> o;
        if (p.is_inf()) {
            o = {F::zero(), F::one(), F::zero()};
        } else {
            Affine<F> a = Affine<F>::canon(XYZZ<F>::to_affine(p));
            o = {a.x, a.y, F::one()};
        }
        store16(q, o);
    }
    MSM_D static PtEl add(const PtEl& a, const PtEl& b) { PtEl r; XYZZ<F>::add_to(r.p, a.p, b.p); return r; }
    MSM_D static PtEl sub(const PtEl& a, const PtEl& b) { PtEl r; const XYZZ<F> nb = XYZZ<F>::neg(b.p); XYZZ<F>::add_to(r.p, a.p, nb); return r; }
    MSM_D static PtEl times(const PtEl& a, const uint32_t* w_mont) { return {g_times_fr<F>(a.p, w_mont)}; }
};
enum { FFT_OP_JOIN = 0, FFT_OP_JOIN_EXT = 1, FFT_OP_JOIN_EXT_INV = 2, FFT_OP_PREP_LAGRANGE = 3 };
// key = first || inc || SM || SC (4 x 8 words, Montgomery)
template <class E, int OP>
__global__ void __launch_bounds__(256) k_fft_pair_op(uint8_t* __restrict__ b1, uint8_t* __restrict__ b2, uint32_t n, const uint32_t* __restrict__ key) {
    using P = typename E::Params;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8], sm[8], sc[8];
    fr_key<P>(key, key + 8, i, w);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sm[k] = key[16 + k];
        sc[k] = key[24 + k];
    }
    E a = E::load(b1 + (size_t)i * E::BYTES), b = E::load(b2 + (size_t)i * E::BYTES);
    if constexpr (OP == FFT_OP_JOIN) {                    // T = W b; (a, b) <- (a + T, a - T)                       :657-746
        const E t = E::times(b, w);
        const E u = a;
        a = E::add(u, t);
        b = E::sub(u, t);
    } else if constexpr (OP == FFT_OP_JOIN_EXT) {          // U = a + b; b <- W (a + SM b); a <- U                     :749-857
        const E u = E::add(a, b);
        b = E::times(E::add(a, E::times(b, sm)), w);
        a = u;
    } else if constexpr (OP == FFT_OP_JOIN_EXT_INV) {      // U = W b; b <- SC (a - U); a <- SC (U - SM a)             :860-988
        const E u = E::times(b, w);
        b = E::times(E::sub(a, u), sc);
        a = E::times(E::sub(u, E::times(a, sm)), sc);
    } else {                                               // U = b - SM a; b <- W (a - b); a <- SC U                  :991-1112
        const E u = E::sub(b, E::times(a, sm));
        b = E::times(E::sub(a, b), w);
        a = E::times(u, sc);
    }
    a.store(b1 + (size_t)i * E::BYTES);
    b.store(b2 + (size_t)i * E::BYTES);
}
// one radix-2 pass of span 2^exp over the whole buffer (every group of 2^exp elements; twiddle omega_(2^exp)^j from the table)
template <class E>
__global__ void __launch_bounds__(256) k_fft_mix(uint8_t* __restrict__ buf, uint32_t n, uint32_t exp, const uint32_t* __restrict__ tw) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t h = 1u << (exp - 1), groups = n >> exp;
    if (idx >= groups * h) return;
    const uint32_t j = idx & (h - 1), i1 = ((idx >> (exp - 1)) << exp) + j, i2 = i1 + h;
    uint32_t w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w[k] = tw[(size_t)j * 8 + k];
    const E u = E::load(buf + (size_t)i1 * E::BYTES);
    const E t = E::times(E::load(buf + (size_t)i2 * E::BYTES), w);
    E::add(u, t).store(buf + (size_t)i1 * E::BYTES);
    E::sub(u, t).store(buf + (size_t)i2 * E::BYTES);
}
// out[i] = factor * in[n - 1 - i]  (in != out)
template <class E>
__global__ void __launch_bounds__(256) k_fft_final(const uint8_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ factor, uint8_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = factor[k];
    E::times(E::load(in + (size_t)(n - 1 - i) * E::BYTES), f).store(out + (size_t)i * E::BYTES);
}
// frm_batchToMontgomery / frm_batchFromMontgomery (build_batchconvertion.js over frm_toMontgomery = mul by R^2, frm_fromMontgomery = mul by 1)
template <class P>
__global__ void __launch_bounds__(256) k_fr_batch_mont(const uint8_t* __restrict__ in, uint32_t n, uint32_t to, uint8_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (to) {
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = P::FR_R2[k];
    }
    FrEl<P>::times(FrEl<P>::load(in + (size_t)i * 32), m).store(out + (size_t)i * 32);
}

}  // namespace msm

// ---- batchApplyKey ----------------------------------------------------------------------------
template <class F>
static int apply_key_t(msm_amd_ctx* ctx, uint32_t flags, const void* in, uint32_t n, const uint8_t* first, const uint8_t* inc, void* out) {
    const bool jac = (flags & MSM_AMD_BASES_JACOBIAN) != 0;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t in_b = (size_t)n * (jac ? sizeof(Jacobian<F>) : sizeof(Affine<F>)), out_b = (size_t)n * sizeof(Jacobian<F>);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = jac ? msm_guard(ctx, 0, &k_batch_apply_key<F, true>, "batch_apply_key") : msm_guard(ctx, 0, &k_batch_apply_key<F, false>, "batch_apply_key")) return rc;
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out, *dkey = nullptr;
    if (dev && (((uintptr_t)in & 15) || ((uintptr_t)out & 15))) {
        msm_set_err(ctx, "device pointers must be 16-byte aligned");
        return MSM_AMD_EINVAL;
    }
    HIPCHK(ctx, hipMalloc(&dkey, 64));
    hipError_t e = hipSuccess;
    if (!dev) {
        void *a = nullptr, *b = nullptr;
        if (hipMalloc(&a, in_b) != hipSuccess || hipMalloc(&b, out_b) != hipSuccess) {
            if (a) (void)hipFree(a);
            (void)hipFree(dkey);
            return msm_alloc_failed(ctx, "batchApplyKey");
        }
        di = a;
        dout = b;
        e = hipMemcpyAsync(di, in, in_b, hipMemcpyHostToDevice, st);
    }
    uint8_t key[64];
    memcpy(key, first, 32);
    memcpy(key + 32, inc, 32);
    if (e == hipSuccess) e = hipMemcpyAsync(dkey, key, 64, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        if (jac) k_batch_apply_key<F, true><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, (const uint32_t*)dkey, (uint8_t*)dout);
        else k_batch_apply_key<F, false><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, (const uint32_t*)dkey, (uint8_t*)dout);
        e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, out_b, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dkey);
    if (!dev) {
        (void)hipFree(di);
        (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("batch_apply_key: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}
// ---- FFT ---------------------------------------------------------------------------------------
template <class P>
static hipError_t fft_twiddles(hipStream_t st, uint32_t n, uint32_t bits, void** dtw) {
    const uint32_t count = n >= 2 ? n / 2 : 1;
    hipError_t e = hipMalloc(dtw, (size_t)count * 32);
    if (e != hipSuccess) return e;
    k_fft_twiddles<P><<<(count + 255) / 256, 256, 0, st>>>(bits, count, (uint32_t*)*dtw);
    return hipGetLastError();
}
template <class P>
static int fr_fft_t(msm_amd_ctx* ctx, uint32_t flags, const void* in, uint32_t n, uint32_t bits, void* out) {
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0, inverse = (flags & MSM_AMD_FFT_INVERSE) != 0;
    const size_t bytes = (size_t)n * 32;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out, *dx = nullptr, *dtw = nullptr;
    hipError_t e = hipMalloc(&dx, bytes);
    if (e == hipSuccess && !dev) {
        e = hipMalloc(&di, bytes);
        if (e == hipSuccess) e = hipMalloc(&dout, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(di, in, bytes, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = fft_twiddles<P>(st, n, bits, &dtw);
    if (e == hipSuccess) {
        const uint32_t nb = (n + 255) / 256, hb = (n / 2 + 255) / 256;
        k_fr_fft_bitrev<P><<<nb, 256, 0, st>>>((const uint32_t*)di, n, bits, (uint32_t*)dx);
        for (uint32_t s = 1; s <= bits; ++s) k_fr_fft_stage<P><<<hb, 256, 0, st>>>((uint32_t*)dx, (const uint32_t*)dtw, n, bits, s);
        k_fr_fft_final<P><<<nb, 256, 0, st>>>((const uint32_t*)dx, n, bits, inverse ? 1u : 0u, (uint32_t*)dout);
        e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (dx) (void)hipFree(dx);
    if (dtw) (void)hipFree(dtw);
    if (!dev) {
        if (di != in) (void)hipFree(di);
        if (dout != out) (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("fft: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}
template <class F>
static int g_fft_t(msm_amd_ctx* ctx, uint32_t flags, const void* in, uint32_t n, uint32_t bits, void* out) {
    using P = typename F::Params;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0, inverse = (flags & MSM_AMD_FFT_INVERSE) != 0;
    const size_t bytes = (size_t)n * sizeof(Jacobian<F>);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = msm_guard(ctx, 0, &k_g_fft_stage<F>, "group fft")) return rc;
    if (int rc = msm_guard(ctx, 0, &k_g_fft_final<F>, "group fft")) return rc;
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out, *dx = nullptr, *dtw = nullptr;
    if (dev && (((uintptr_t)in & 15) || ((uintptr_t)out & 15))) {
        msm_set_err(ctx, "device pointers must be 16-byte aligned");
        return MSM_AMD_EINVAL;
    }
    hipError_t e = hipMalloc(&dx, (size_t)n * sizeof(XYZZ<F>));
    if (e == hipSuccess && !dev) {
        e = hipMalloc(&di, bytes);
        if (e == hipSuccess) e = hipMalloc(&dout, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(di, in, bytes, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) e = fft_twiddles<P>(st, n, bits, &dtw);
    if (e == hipSuccess) {
        const uint32_t nb = (n + 255) / 256, hb = (n / 2 + 255) / 256;
        k_g_fft_bitrev<F><<<nb, 256, 0, st>>>((const uint8_t*)di, n, bits, (uint8_t*)dx);
        for (uint32_t s = 1; s <= bits; ++s) k_g_fft_stage<F><<<hb, 256, 0, st>>>((uint8_t*)dx, (const uint32_t*)dtw, n, bits, s);
        k_g_fft_final<F><<<nb, 256, 0, st>>>((const uint8_t*)dx, n, bits, inverse ? 1u : 0u, (uint8_t*)dout);
        e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (dx) (void)hipFree(dx);
    if (dtw) (void)hipFree(dtw);
    if (!dev) {
        if (di != in) (void)hipFree(di);
        if (dout != out) (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("fft: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}

// ---- blocked-FFT helpers ------------------------------------------------------------------------
// which: 0..3 = pair operations (FFT_OP_*), 4 = fftMix (arg = exp), 5 = fftFinal (key32 = factor).  Buffers are transformed in place.
template <class E>
static int fft_helper_t(msm_amd_ctx* ctx, int which, uint32_t flags, void* b1, void* b2, uint32_t n, const uint8_t* first32, const uint8_t* inc32,
                        uint32_t arg) {
    using P = typename E::Params;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t bytes = (size_t)n * E::BYTES;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {   // (point elements: out-of-line point arithmetic, scratch on the queue -- engine_ctx.hpp; the Fr forms need none)
        int rc = 0;
        if (which == FFT_OP_JOIN) rc = msm_guard(ctx, 0, &k_fft_pair_op<E, FFT_OP_JOIN>, "fft_join");
        else if (which == FFT_OP_JOIN_EXT) rc = msm_guard(ctx, 0, &k_fft_pair_op<E, FFT_OP_JOIN_EXT>, "fft_join_ext");
        else if (which == FFT_OP_JOIN_EXT_INV) rc = msm_guard(ctx, 0, &k_fft_pair_op<E, FFT_OP_JOIN_EXT_INV>, "fft_join_ext");
        else if (which == FFT_OP_PREP_LAGRANGE) rc = msm_guard(ctx, 0, &k_fft_pair_op<E, FFT_OP_PREP_LAGRANGE>, "prepare_lagrange_evaluation");
        else if (which == 4) rc = msm_guard(ctx, 0, &k_fft_mix<E>, "fft_mix");
        else rc = msm_guard(ctx, 0, &k_fft_final<E>, "fft_final");
        if (rc) return rc;
    }
    hipStream_t st = ctx->stream;
    if (dev && (((uintptr_t)b1 & 15) || ((uintptr_t)b2 & 15))) {
        msm_set_err(ctx, "device pointers must be 16-byte aligned");
        return MSM_AMD_EINVAL;
    }
    const bool pair = which <= 3;
    if ((pair || which == 4) && arg > (uint32_t)P::FR_MAXBITS) {
        msm_set_err(ctx, "totalBits / exp beyond the 2-adicity of the scalar field");
        return MSM_AMD_EINVAL;
    }
    if (which == 4 && (arg == 0 || (n & ((1u << arg) - 1)) != 0)) {
        msm_set_err(ctx, "fftMix: n must be a multiple of 2^exp, exp >= 1");
        return MSM_AMD_EINVAL;
    }
    void *d1 = b1, *d2 = b2, *dkey = nullptr, *dtw = nullptr, *dtmp = nullptr;
    hipError_t e = hipMalloc(&dkey, 128);
    if (e == hipSuccess && !dev) {
        e = hipMalloc(&d1, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(d1, b1, bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && pair) {
            e = hipMalloc(&d2, bytes);
            if (e == hipSuccess) e = hipMemcpyAsync(d2, b2, bytes, hipMemcpyHostToDevice, st);
        }
    }
    uint8_t key[128];
    memset(key, 0, sizeof key);
    if (first32) memcpy(key, first32, 32);
    if (inc32) memcpy(key + 32, inc32, 32);
    if (pair) {
        memcpy(key + 64, P::FR_SHIFT_TO_M[arg], 32);
        memcpy(key + 96, P::FR_SCONST[arg], 32);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dkey, key, 128, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const uint32_t nb = (n + 255) / 256;
        uint8_t *p1 = (uint8_t*)d1, *p2 = (uint8_t*)d2;
        const uint32_t* k = (const uint32_t*)dkey;
        if (which == FFT_OP_JOIN) k_fft_pair_op<E, FFT_OP_JOIN><<<nb, 256, 0, st>>>(p1, p2, n, k);
        else if (which == FFT_OP_JOIN_EXT) k_fft_pair_op<E, FFT_OP_JOIN_EXT><<<nb, 256, 0, st>>>(p1, p2, n, k);
        else if (which == FFT_OP_JOIN_EXT_INV) k_fft_pair_op<E, FFT_OP_JOIN_EXT_INV><<<nb, 256, 0, st>>>(p1, p2, n, k);
        else if (which == FFT_OP_PREP_LAGRANGE) k_fft_pair_op<E, FFT_OP_PREP_LAGRANGE><<<nb, 256, 0, st>>>(p1, p2, n, k);
        else if (which == 4) {
            e = fft_twiddles<P>(st, 1u << arg, arg, &dtw);
            if (e == hipSuccess) k_fft_mix<E><<<(n / 2 + 255) / 256, 256, 0, st>>>(p1, n, arg, (const uint32_t*)dtw);
        } else {
            e = hipMalloc(&dtmp, bytes);
            if (e == hipSuccess) {
                k_fft_final<E><<<nb, 256, 0, st>>>(p1, n, k, (uint8_t*)dtmp);
                e = hipMemcpyAsync(d1, dtmp, bytes, hipMemcpyDeviceToDevice, st);
            }
        }
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) {
        e = hipMemcpyAsync(b1, d1, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && pair) e = hipMemcpyAsync(b2, d2, bytes, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    else (void)hipStreamSynchronize(st);
    if (dkey) (void)hipFree(dkey);
    if (dtw) (void)hipFree(dtw);
    if (dtmp) (void)hipFree(dtmp);
    if (!dev) {
        if (d1 && d1 != b1) (void)hipFree(d1);
        if (d2 && d2 != b2) (void)hipFree(d2);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("fft helper: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}
static int fft_helper(msm_amd_ctx* ctx, int which, int curve, int group, uint32_t flags, void* b1, void* b2, uint32_t n, const uint8_t* first32,
                      const uint8_t* inc32, uint32_t arg) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!b1 || (which <= 3 && (!b2 || !first32 || !inc32)) || (which == 5 && !first32)) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (curve == MSM_AMD_BLS12381) {
        using P = Bls12381Fp;
        if (group == 0) return fft_helper_t<FrEl<P>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
        if (group == 1) return fft_helper_t<PtEl<Fp<P>>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
        if (group == 2) return fft_helper_t<PtEl<Fp2<P>>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
    } else if (curve == MSM_AMD_BN128) {
        using P = Bn254Fp;
        if (group == 0) return fft_helper_t<FrEl<P>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
        if (group == 1) return fft_helper_t<PtEl<Fp<P>>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
        if (group == 2) return fft_helper_t<PtEl<Fp2<P>>>(ctx, which, flags, b1, b2, n, first32, inc32, arg);
    }
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}

extern "C" {

int msm_amd_fft(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* in, uint32_t n, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if ((n & (n - 1)) != 0 || n > (1u << 27)) {
        msm_set_err(ctx, "fft: n must be a power of two, at most 2^27");
        return MSM_AMD_EINVAL;
    }
    uint32_t bits = 0;
    while ((1u << bits) < n) ++bits;
    if (curve == MSM_AMD_BLS12381) {
        using P = Bls12381Fp;
        if (group == 0) return fr_fft_t<P>(ctx, flags, in, n, bits, out);
        if (group == 1) return g_fft_t<Fp<P>>(ctx, flags, in, n, bits, out);
        if (group == 2) return g_fft_t<Fp2<P>>(ctx, flags, in, n, bits, out);
    } else if (curve == MSM_AMD_BN128) {
        using P = Bn254Fp;
        if (group == 0) return fr_fft_t<P>(ctx, flags, in, n, bits, out);
        if (group == 1) return g_fft_t<Fp<P>>(ctx, flags, in, n, bits, out);
        if (group == 2) return g_fft_t<Fp2<P>>(ctx, flags, in, n, bits, out);
    }
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}
int msm_amd_batch_apply_key(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* in, uint32_t n, const uint8_t* first32,
                            const uint8_t* inc32, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out || !first32 || !inc32) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (curve == MSM_AMD_BLS12381 && group == 1) return apply_key_t<Fp<Bls12381Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BLS12381 && group == 2) return apply_key_t<Fp2<Bls12381Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BN128 && group == 1) return apply_key_t<Fp<Bn254Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BN128 && group == 2) return apply_key_t<Fp2<Bn254Fp>>(ctx, flags, in, n, first32, inc32, out);
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}
int msm_amd_fr_batch_apply_key(msm_amd_ctx* ctx, int curve, uint32_t flags, const void* in, uint32_t n, const uint8_t* first32, const uint8_t* inc32,
                               void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out || !first32 || !inc32 || (curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128)) {
        msm_set_err(ctx, "bad argument");
        return MSM_AMD_EINVAL;
    }
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t bytes = (size_t)n * 32;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out, *dkey = nullptr;
    HIPCHK(ctx, hipMalloc(&dkey, 64));
    hipError_t e = hipSuccess;
    if (!dev) {
        if (hipMalloc(&di, bytes) != hipSuccess || hipMalloc(&dout, bytes) != hipSuccess) {
            msm_set_err(ctx, "hipMalloc failed");
            return MSM_AMD_EHIP;
        }
        e = hipMemcpyAsync(di, in, bytes, hipMemcpyHostToDevice, st);
    }
    uint8_t key[64];
    memcpy(key, first32, 32);
    memcpy(key + 32, inc32, 32);
    if (e == hipSuccess) e = hipMemcpyAsync(dkey, key, 64, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        if (curve == MSM_AMD_BLS12381) k_fr_batch_apply_key<Bls12381Fp><<<(n + 255) / 256, 256, 0, st>>>((const uint32_t*)di, n, (const uint32_t*)dkey, (uint32_t*)dout);
        else k_fr_batch_apply_key<Bn254Fp><<<(n + 255) / 256, 256, 0, st>>>((const uint32_t*)di, n, (const uint32_t*)dkey, (uint32_t*)dout);
        e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dkey);
    if (!dev) {
        (void)hipFree(di);
        (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("fr_batch_apply_key: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}


int msm_amd_fft_join(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, void* buff1, void* buff2, uint32_t n, const uint8_t* first32,
                     const uint8_t* inc32) {
    return fft_helper(ctx, FFT_OP_JOIN, curve, group, flags, buff1, buff2, n, first32, inc32, 0);
}
int msm_amd_fft_join_ext(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, void* buff1, void* buff2, uint32_t n, const uint8_t* first32,
                         const uint8_t* inc32, uint32_t total_bits) {
    return fft_helper(ctx, (flags & MSM_AMD_FFT_INVERSE) ? FFT_OP_JOIN_EXT_INV : FFT_OP_JOIN_EXT, curve, group, flags, buff1, buff2, n, first32, inc32,
                      total_bits);
}
int msm_amd_prepare_lagrange_evaluation(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, void* buff1, void* buff2, uint32_t n,
                                        const uint8_t* first32, const uint8_t* inc32, uint32_t total_bits) {
    return fft_helper(ctx, FFT_OP_PREP_LAGRANGE, curve, group, flags, buff1, buff2, n, first32, inc32, total_bits);
}
int msm_amd_fft_mix(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, void* buff, uint32_t n, uint32_t exp) {
    return fft_helper(ctx, 4, curve, group, flags, buff, nullptr, n, nullptr, nullptr, exp);
}
int msm_amd_fft_final(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, void* buff, uint32_t n, const uint8_t* factor32) {
    return fft_helper(ctx, 5, curve, group, flags, buff, nullptr, n, factor32, nullptr, 0);
}
int msm_amd_fr_batch_montgomery(msm_amd_ctx* ctx, int curve, uint32_t flags, const void* in, uint32_t n, int to_montgomery, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out || (curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128)) {
        msm_set_err(ctx, "bad argument");
        return MSM_AMD_EINVAL;
    }
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t bytes = (size_t)n * 32;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out;
    hipError_t e = hipSuccess;
    if (!dev) {
        e = hipMalloc(&di, bytes);
        if (e == hipSuccess) e = hipMalloc(&dout, bytes);
        if (e == hipSuccess) e = hipMemcpyAsync(di, in, bytes, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) {
        if (curve == MSM_AMD_BLS12381) k_fr_batch_mont<Bls12381Fp><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, to_montgomery ? 1u : 0u, (uint8_t*)dout);
        else k_fr_batch_mont<Bn254Fp><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, to_montgomery ? 1u : 0u, (uint8_t*)dout);
        e = hipGetLastError();
    }
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (!dev) {
        if (di != in) (void)hipFree(di);
        if (dout != out) (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("fr_batch_montgomery: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}

}  // extern "C"
