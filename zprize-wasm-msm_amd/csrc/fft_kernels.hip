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

}  // namespace msm

// ---- batchApplyKey ----------------------------------------------------------------------------
template <class F>
static int apply_key_t(msm_amd_ctx* ctx, uint32_t flags, const void* in, uint32_t n, const uint8_t* first, const uint8_t* inc, void* out) {
    const bool jac = (flags & MSM_AMD_BASES_JACOBIAN) != 0;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t in_b = (size_t)n * (jac ? sizeof(Jacobian<F>) : sizeof(Affine<F>)), out_b = (size_t)n * sizeof(Jacobian<F>);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *di = const_cast<void*>(in), *dout = out, *dkey = nullptr;
    if (dev && (((uintptr_t)in & 15) || ((uintptr_t)out & 15))) {
        msm_set_err(ctx, "device pointers must be 16-byte aligned");
        return MSM_AMD_EINVAL;
    }
    HIPCHK(ctx, hipMalloc(&dkey, 64));
    hipError_t e = hipSuccess;
    if (!dev) {
        if (hipMalloc(&di, in_b) != hipSuccess || hipMalloc(&dout, out_b) != hipSuccess) {
            msm_set_err(ctx, "hipMalloc failed");
            return MSM_AMD_EHIP;
        }
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

}  // extern "C"
