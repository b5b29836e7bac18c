// conv_kernels.hip -- the data formats either side of the MSM (SURVEY.md 8f rank 2 and the GLV pre-pass of rank 1): batch
// conversions, compressed forms, on-curve / subgroup checks, the reference's g1m_glv_* pre-pass, synthetic base generation.
// One lane per point; none of this is on the MSM's own hot path.
#include "engine_ctx.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------
// Batch conversions either side of the MSM, one lane per point:
//   k_batch_to_jacobian   g?m_batchToJacobian  (build_curve_jacobian_a0.js:1014-1038,1418)
//   k_batch_lem_to_u      g?m_batchLEMtoU      (:1206-1231,1413)  LE Montgomery affine -> x||y big-endian plain, inf = 0x40 00..
//   k_batch_u_to_lem      g?m_batchUtoLEM      (:1233-1258,1415)
// (g?m_batchToAffine is k_jacobian_to_affine above.)
// ------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_batch_to_jacobian(const uint8_t* __restrict__ aff, uint32_t n, uint8_t* __restrict__ jac) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = load16<Affine<F>>(aff + (size_t)i * sizeof(Affine<F>));
    Jacobian<F> j;
    if (p.is_inf()) j = {F::zero(), F::one(), F::zero()};
    else j = {p.x, p.y, F::one()};
    store16(jac + (size_t)i * sizeof(Jacobian<F>), j);
}
MSM_D uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
template <class F, class B>
__global__ void __launch_bounds__(256) k_batch_lem_to_u(const uint8_t* __restrict__ aff, uint32_t n, uint8_t* __restrict__ out) {
    constexpr int T = sizeof(F) / 4, NB = sizeof(B) / 4, COMP = T / NB;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = load16<Affine<F>>(aff + (size_t)i * sizeof(Affine<F>));
    Affine<F> o;
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
    if (p.is_inf()) {
#pragma unroll
        for (int k = 0; k < 2 * T; ++k) ow[k] = 0;
        ow[0] = 0x40u;   // first byte of the output
    } else {
        const F* coords[2] = {&p.x, &p.y};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t plain[T];
            const B* comps = reinterpret_cast<const B*>(coords[c]);
#pragma unroll
            for (int k = 0; k < COMP; ++k) {
                B v = B::canon(B::from_mont(comps[k]));
#pragma unroll
                for (int l = 0; l < NB; ++l) plain[k * NB + l] = v.v[l];
            }
#pragma unroll
            for (int l = 0; l < T; ++l) ow[c * T + l] = bswap32(plain[T - 1 - l]);
        }
    }
    store16(out + (size_t)i * sizeof(Affine<F>), o);
}
template <class F, class B>
__global__ void __launch_bounds__(256) k_batch_u_to_lem(const uint8_t* __restrict__ in, uint32_t n, uint8_t* __restrict__ aff) {
    constexpr int T = sizeof(F) / 4, NB = sizeof(B) / 4, COMP = T / NB;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> u = load16<Affine<F>>(in + (size_t)i * sizeof(Affine<F>));
    const uint32_t* uw = reinterpret_cast<const uint32_t*>(&u);
    Affine<F> o;
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
    if (uw[0] & 0x40u) {
#pragma unroll
        for (int k = 0; k < 2 * T; ++k) ow[k] = 0;
    } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int k = 0; k < COMP; ++k) {
                B v;
#pragma unroll
                for (int l = 0; l < NB; ++l) v.v[l] = bswap32(uw[c * T + (T - 1 - (k * NB + l))]);
                B m = B::canon(B::to_mont(v));
#pragma unroll
                for (int l = 0; l < NB; ++l) ow[c * T + k * NB + l] = m.v[l];
            }
        }
    }
    store16(aff + (size_t)i * sizeof(Affine<F>), o);
}


// g?m_inCurveAffine (build_curve_jacobian_a0.js:1330-1353) for a batch: flag[i] = (y^2 == x^3 + b).
// As in the reference, the affine infinity (0, 0) is not "in curve" (0 != b).
template <class F>
__global__ void __launch_bounds__(256) k_batch_in_curve_affine(const uint8_t* __restrict__ aff, uint32_t n, F b, uint8_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = load16<Affine<F>>(aff + (size_t)i * sizeof(Affine<F>));
    F y2 = F::sqr(p.y);
    F x3b = F::add(F::mul(p.x, F::sqr(p.x)), b);
    flags[i] = F::sub(y2, x3b).is_zero() ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// Compressed point form: g?m_LEMtoC / g?m_CtoLEM, build_curve_jacobian_a0.js:1168-1205, :1265-1328;
// sign convention build_f1m.js:135-156 (plain value >= (q+1)/2 is "negative"), build_f2m.js:411-430 (c1 first, then c0).
// LEMtoC: x big-endian plain, 0x80 on the first byte if sign(y) == -1; infinity (x = y = 0) = 0x40 00 .. 00 -- the reference
// tests infinity with the Jacobian _isZero, i.e. it looks past its affine input; this is the intended reading.
// CtoLEM: y = sqrt(x^3 + b) with the sign the 0x80 flag asks for.  Both base fields have q = 3 mod 4: sqrt = a^((q+1)/4);
// in Fp2 = Fp[u]/(u^2+1) the complex method.  Whichever root the reference's Tonelli-Shanks returns, the sign fix makes y unique.
// ------------------------------------------------------------------------------------------
template <class B>
MSM_D B fp_pow_limbs(const B& base, const uint32_t* e) {
    B acc = B::one(), b = base;
    for (int i = 0; i < 32 * B::N; ++i) {
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = B::mul(acc, b);
        b = B::sqr(b);
    }
    return acc;
}
template <class B>
MSM_D int fp_sign_of(const B& x) {
    using P = typename B::Params;
    if (x.is_zero()) return 0;
    B pl = B::canon(B::from_mont(x));
    for (int i = B::N - 1; i >= 0; --i) {
        if (pl.v[i] > P::QP1D2[i]) return -1;
        if (pl.v[i] < P::QP1D2[i]) return 1;
    }
    return -1;
}
template <class B>
MSM_D bool fp_sqrt_of(const B& a, B& r) {
    using P = typename B::Params;
    r = fp_pow_limbs<B>(a, P::QP1D4);
    return B::eq(B::sqr(r), a);
}
template <class F>
struct CompOps;
template <class P>
struct CompOps<Fp<P>> {
    using F = Fp<P>;
    MSM_D static int sign(const F& x) { return fp_sign_of<F>(x); }
    MSM_D static bool sqrt(const F& a, F& r) { return fp_sqrt_of<F>(a, r); }
};
template <class P>
struct CompOps<Fp2<P>> {
    using F = Fp2<P>;
    using B = Fp<P>;
    MSM_D static int sign(const F& x) {
        int s = fp_sign_of<B>(x.c1);
        return s ? s : fp_sign_of<B>(x.c0);
    }
    MSM_D static bool sqrt(const F& a, F& r) {
        if (a.c1.is_zero()) {
            if (fp_sqrt_of<B>(a.c0, r.c0)) {
                r.c1 = B::zero();
                return true;
            }
            r.c0 = B::zero();
            return fp_sqrt_of<B>(B::neg(a.c0), r.c1);
        }
        B alpha;
        if (!fp_sqrt_of<B>(B::add(B::sqr(a.c0), B::sqr(a.c1)), alpha)) return false;
        const B inv2 = B::inv(B::add(B::one(), B::one()));
        B x0;
        if (!fp_sqrt_of<B>(B::mul(B::add(a.c0, alpha), inv2), x0))
            if (!fp_sqrt_of<B>(B::mul(B::sub(a.c0, alpha), inv2), x0)) return false;
        r.c0 = x0;
        r.c1 = B::mul(a.c1, B::inv(B::add(x0, x0)));
        return F::eq(F::sqr(r), a);
    }
};
template <class F, class B>
__global__ void __launch_bounds__(256) k_batch_lem_to_c(const uint8_t* __restrict__ aff, uint32_t n, uint8_t* __restrict__ out) {
    constexpr int T = sizeof(F) / 4, NB = sizeof(B) / 4, COMP = T / NB;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = load16<Affine<F>>(aff + (size_t)i * sizeof(Affine<F>));
    F o;
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
    if (p.is_inf()) {
#pragma unroll
        for (int k = 0; k < T; ++k) ow[k] = 0;
        ow[0] = 0x40u;
    } else {
        uint32_t plain[T];
        const B* comps = reinterpret_cast<const B*>(&p.x);
#pragma unroll
        for (int k = 0; k < COMP; ++k) {
            B v = B::canon(B::from_mont(comps[k]));
#pragma unroll
            for (int l = 0; l < NB; ++l) plain[k * NB + l] = v.v[l];
        }
#pragma unroll
        for (int l = 0; l < T; ++l) ow[l] = bswap32(plain[T - 1 - l]);
        if (CompOps<F>::sign(p.y) == -1) ow[0] |= 0x80u;
    }
    store16(out + (size_t)i * sizeof(F), o);
}
template <class F, class B>
__global__ void __launch_bounds__(256) k_batch_c_to_lem(const uint8_t* __restrict__ in, uint32_t n, F b, uint8_t* __restrict__ aff, uint32_t* __restrict__ bad) {
    constexpr int T = sizeof(F) / 4, NB = sizeof(B) / 4, COMP = T / NB;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F u = load16<F>(in + (size_t)i * sizeof(F));
    uint32_t* uw = reinterpret_cast<uint32_t*>(&u);
    Affine<F> o;
    uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
    if (uw[0] & 0x40u) {
#pragma unroll
        for (int k = 0; k < 2 * T; ++k) ow[k] = 0;
    } else {
        const bool greatest = (uw[0] & 0x80u) != 0;
        uw[0] &= ~0xC0u;
        uint32_t plain[T];
#pragma unroll
        for (int l = 0; l < T; ++l) plain[l] = bswap32(uw[T - 1 - l]);
        B* xc = reinterpret_cast<B*>(&o.x);
#pragma unroll
        for (int k = 0; k < COMP; ++k) {
            B v;
#pragma unroll
            for (int l = 0; l < NB; ++l) v.v[l] = plain[k * NB + l];
            xc[k] = B::canon(B::to_mont(v));
        }
        F y;
        if (!CompOps<F>::sqrt(F::add(F::mul(o.x, F::sqr(o.x)), b), y)) {
            atomicAdd(bad, 1u);
            y = F::zero();
        }
        y = F::canon(y);
        if ((CompOps<F>::sign(y) == -1) != greatest) y = F::canon(F::neg(y));
        o.y = y;
    }
    store16(aff + (size_t)i * sizeof(Affine<F>), o);
}

// bls12381/build_bls12381.js:1494-1576 / :1370-1480 (g?m_inGroupAffine), batched: 1 for affine infinity, 0 off the curve, else
// r*P == infinity (the definition; the reference reaches the same predicate through endomorphism identities).  One lane per
// point, 255 doublings + ~128 mixed additions; a validation pass that runs once per base table.
template <class F>
__global__ void __launch_bounds__(256) k_batch_in_group_affine(const uint8_t* __restrict__ aff, uint32_t n, F b, uint8_t* __restrict__ flags) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = load16<Affine<F>>(aff + (size_t)i * sizeof(Affine<F>));
    if (p.is_inf()) {
        flags[i] = 1;
        return;
    }
    F y2 = F::sqr(p.y);
    F x3b = F::add(F::mul(p.x, F::sqr(p.x)), b);
    if (!F::sub(y2, x3b).is_zero()) {
        flags[i] = 0;
        return;
    }
    using P = typename F::Params;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 254; bit >= 0; --bit) {
        acc = XYZZ<F>::dbl_ni(acc);
        if ((P::RORDER[bit >> 5] >> (bit & 31)) & 1u) XYZZ<F>::madd(acc, p.x, p.y);
    }
    flags[i] = acc.is_inf() ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// GLV pre-pass (BLS12-381 G1): build_glv.js:178-267 (_preprocessEndomorphism), one lane per pair.
//   points_out[2i]   = (x, k1 >= 0 ? y : -y)        scalars_out[2i]   = |k1|  (32 bytes, upper 16 zero)
//   points_out[2i+1] = (beta*x, k2 >= 0 ? y : -y)   scalars_out[2i+1] = |k2|
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_glv_preprocess(const uint8_t* __restrict__ points, const uint32_t* __restrict__ scalars, uint32_t n,
                                                         uint8_t* __restrict__ points_out, uint32_t* __restrict__ scalars_out) {
    using F = Fp<Bls12381Fp>;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = scalars[(size_t)i * 8 + j];
    uint32_t sign = glv::decompose(k, k1, k2);
    Affine<F> p = load16<Affine<F>>(points + (size_t)i * sizeof(Affine<F>));
    F beta;
#pragma unroll
    for (int j = 0; j < 12; ++j) beta.v[j] = glv::Consts::BETA[j];
    Affine<F> a{p.x, F::cneg(p.y, (sign & 1u) == 0)};
    Affine<F> b{F::mul(p.x, beta), F::cneg(p.y, (sign & 2u) == 0)};
    store16(points_out + (size_t)(2 * i) * sizeof(Affine<F>), Affine<F>::canon(a));
    store16(points_out + (size_t)(2 * i + 1) * sizeof(Affine<F>), Affine<F>::canon(b));
    uint32_t* so = scalars_out + (size_t)i * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        so[j] = k1[j];
        so[4 + j] = 0;
        so[8 + j] = k2[j];
        so[12 + j] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// synthetic bases for benchmarks/tests: P_i = k_i * G, k_i from splitmix64(seed, i)
// ------------------------------------------------------------------------------------------
MSM_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
template <class F>
__global__ void __launch_bounds__(256) k_gen_points(Affine<F> gen, uint64_t seed, uint32_t first, uint32_t n, uint8_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k[4];   // the multiplier of point first + i of the sequence (a slice of it can be generated on its own)
    for (int j = 0; j < 4; ++j) k[j] = splitmix64(seed * 0x100000001B3ull + ((uint64_t)first + i) * 4 + j);
    k[3] &= 0x3fffffffffffffffull;   // 254-bit multipliers
    if ((k[0] | k[1] | k[2] | k[3]) == 0) k[0] = 1;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 253; bit >= 0; --bit) {
        acc = XYZZ<F>::dbl(acc);
        if ((k[bit >> 6] >> (bit & 63)) & 1ull) XYZZ<F>::madd(acc, gen.x, gen.y);
    }
    Affine<F> a = Affine<F>::canon(XYZZ<F>::to_affine(acc));
    store16(out + (size_t)i * sizeof(Affine<F>), a);
}

}  // namespace msm

template <class F>
static int gen_points_t(msm_amd_ctx* ctx, const Affine<F>& gen, uint64_t seed, uint32_t first, uint32_t n, void* dev_out) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (int rc = msm_guard(ctx, 0, &k_gen_points<F>, "gen_points")) return rc;
    k_gen_points<F><<<(n + 255) / 256, 256, 0, ctx->stream>>>(gen, seed, first, n, (uint8_t*)dev_out);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// ---- batch conversions (g?m_batchToAffine / _batchToJacobian / _batchLEMtoU / _batchUtoLEM) ----------------
template <class F, class B>
static int conv_t(msm_amd_ctx* ctx, int which, uint32_t flags, const void* in, uint32_t n, void* out, const F* curve_b = nullptr) {
    const size_t ASZ = sizeof(Affine<F>), JSZ = sizeof(Jacobian<F>), FSZ = sizeof(F);
    const size_t in_b = (size_t)n * (which == 0 ? JSZ : which == 5 ? FSZ : ASZ), out_b = (size_t)n * (which == 1 ? JSZ : which == 4 ? FSZ : ASZ);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    void *di = const_cast<void*>(in), *dout = out;
    if (dev) {
        if (((uintptr_t)in & 15) || ((uintptr_t)out & 15)) {
            msm_set_err(ctx, "device pointers must be 16-byte aligned");
            return MSM_AMD_EINVAL;
        }
    } else {
        HIPCHK(ctx, hipMalloc(&di, in_b));
        if (hipMalloc(&dout, out_b) != hipSuccess) {
            (void)hipFree(di);
            return msm_alloc_failed(ctx, "batch conversion");
        }
        (void)hipMemcpyAsync(di, in, in_b, hipMemcpyHostToDevice, st);
    }
    const uint32_t blocks = (n + 255) / 256;
    if (which == 0) k_jacobian_to_affine<F><<<((n + jacobian_to_affine_points_per_lane<F>() - 1) / jacobian_to_affine_points_per_lane<F>() + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
    else if (which == 1) k_batch_to_jacobian<F><<<blocks, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
    else if (which == 2) k_batch_lem_to_u<F, B><<<blocks, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
    else if (which == 3) k_batch_u_to_lem<F, B><<<blocks, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
    else if (which == 4) k_batch_lem_to_c<F, B><<<blocks, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
    uint32_t* dbad = nullptr;
    uint32_t hbad = 0;
    if (which == 5) {
        if (hipMalloc(&dbad, 4) == hipSuccess) {
            (void)hipMemsetAsync(dbad, 0, 4, st);
            k_batch_c_to_lem<F, B><<<blocks, 256, 0, st>>>((const uint8_t*)di, n, *curve_b, (uint8_t*)dout, dbad);
            (void)hipMemcpyAsync(&hbad, dbad, 4, hipMemcpyDeviceToHost, st);
        }
    }
    hipError_t e = hipGetLastError();
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, out_b, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (dbad) (void)hipFree(dbad);
    if (!dev) {
        (void)hipFree(di);
        (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("batch conversion: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    if (hbad) {
        msm_set_err(ctx, "CtoLEM: x^3 + b is not a square for " + std::to_string(hbad) + " element(s) (written as infinity; the reference does not terminate on such input)");
        return MSM_AMD_EINVAL;
    }
    return 0;
}
static int conv(msm_amd_ctx* ctx, int which, int curve, int group, uint32_t flags, const void* in, uint32_t n, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (curve == MSM_AMD_BLS12381 && group == 1) {
        using P = Bls12381Fp;
        const Fp<P> b = fp_from<P>(P::B1);
        return conv_t<Fp<P>, Fp<P>>(ctx, which, flags, in, n, out, &b);
    }
    if (curve == MSM_AMD_BLS12381 && group == 2) {
        using P = Bls12381Fp;
        const Fp2<P> b{fp_from<P>(P::B2_0), fp_from<P>(P::B2_1)};
        return conv_t<Fp2<P>, Fp<P>>(ctx, which, flags, in, n, out, &b);
    }
    if (curve == MSM_AMD_BN128 && group == 1) {
        using P = Bn254Fp;
        const Fp<P> b = fp_from<P>(P::B1);
        return conv_t<Fp<P>, Fp<P>>(ctx, which, flags, in, n, out, &b);
    }
    if (curve == MSM_AMD_BN128 && group == 2) {
        using P = Bn254Fp;
        const Fp2<P> b{fp_from<P>(P::B2_0), fp_from<P>(P::B2_1)};
        return conv_t<Fp2<P>, Fp<P>>(ctx, which, flags, in, n, out, &b);
    }
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}

template <class F>
static int in_curve_t(msm_amd_ctx* ctx, const F& b, uint32_t flags, const void* points, uint32_t n, uint8_t* out, bool group_check = false) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (group_check)   // (its scalar multiplications call out-of-line point arithmetic: scratch on the queue, engine_ctx.hpp)
        if (int rc = msm_guard(ctx, 0, &k_batch_in_group_affine<F>, "batch_in_group_affine")) return rc;
    hipStream_t st = ctx->stream;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t in_b = (size_t)n * sizeof(Affine<F>);
    void *di = const_cast<void*>(points), *dout = out;
    if (dev) {
        if ((uintptr_t)points & 15) {
            msm_set_err(ctx, "device points pointer must be 16-byte aligned");
            return MSM_AMD_EINVAL;
        }
    } else {
        HIPCHK(ctx, hipMalloc(&di, in_b));
        if (hipMalloc(&dout, n) != hipSuccess) {
            (void)hipFree(di);
            return msm_alloc_failed(ctx, "batch check");
        }
        (void)hipMemcpyAsync(di, points, in_b, hipMemcpyHostToDevice, st);
    }
    if (group_check) k_batch_in_group_affine<F><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, b, (uint8_t*)dout);
    else k_batch_in_curve_affine<F><<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, b, (uint8_t*)dout);
    hipError_t e = hipGetLastError();
    if (!dev && e == hipSuccess) e = hipMemcpyAsync(out, dout, n, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (!dev) {
        (void)hipFree(di);
        (void)hipFree(dout);
    }
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("batch_in_curve_affine: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}

extern "C" {

int msm_amd_batch_to_affine(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* jac_in, uint32_t n, void* aff_out) {
    return conv(ctx, 0, curve, group, flags, jac_in, n, aff_out);
}
int msm_amd_batch_to_jacobian(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* aff_in, uint32_t n, void* jac_out) {
    return conv(ctx, 1, curve, group, flags, aff_in, n, jac_out);
}
int msm_amd_batch_lem_to_u(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* aff_in, uint32_t n, void* u_out) {
    return conv(ctx, 2, curve, group, flags, aff_in, n, u_out);
}
int msm_amd_batch_u_to_lem(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* u_in, uint32_t n, void* aff_out) {
    return conv(ctx, 3, curve, group, flags, u_in, n, aff_out);
}
int msm_amd_batch_lem_to_c(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* aff_in, uint32_t n, void* c_out) {
    return conv(ctx, 4, curve, group, flags, aff_in, n, c_out);
}
int msm_amd_batch_c_to_lem(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* c_in, uint32_t n, void* aff_out) {
    return conv(ctx, 5, curve, group, flags, c_in, n, aff_out);
}

int msm_amd_batch_in_group_affine(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* points, uint32_t n, uint8_t* out_flags) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n && (!points || !out_flags)) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (curve != MSM_AMD_BLS12381 || (group != 1 && group != 2)) {
        msm_set_err(ctx, "inGroupAffine exists for BLS12-381 only (as in the reference)");
        return MSM_AMD_EINVAL;
    }
    if (n == 0) return 0;
    using P = Bls12381Fp;
    if (group == 1) return in_curve_t<Fp<P>>(ctx, fp_from<P>(P::B1), flags, points, n, out_flags, true);
    return in_curve_t<Fp2<P>>(ctx, Fp2<P>{fp_from<P>(P::B2_0), fp_from<P>(P::B2_1)}, flags, points, n, out_flags, true);
}
int msm_amd_batch_in_curve_affine(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* points, uint32_t n, uint8_t* out_flags) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!points || !out_flags) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (curve == MSM_AMD_BLS12381 && group == 1) return in_curve_t<Fp<Bls12381Fp>>(ctx, fp_from<Bls12381Fp>(Bls12381Fp::B1), flags, points, n, out_flags);
    if (curve == MSM_AMD_BN128 && group == 1) return in_curve_t<Fp<Bn254Fp>>(ctx, fp_from<Bn254Fp>(Bn254Fp::B1), flags, points, n, out_flags);
    if (curve == MSM_AMD_BLS12381 && group == 2) {
        using P = Bls12381Fp;
        return in_curve_t<Fp2<P>>(ctx, Fp2<P>{fp_from<P>(P::B2_0), fp_from<P>(P::B2_1)}, flags, points, n, out_flags);
    }
    if (curve == MSM_AMD_BN128 && group == 2) {
        using P = Bn254Fp;
        return in_curve_t<Fp2<P>>(ctx, Fp2<P>{fp_from<P>(P::B2_0), fp_from<P>(P::B2_1)}, flags, points, n, out_flags);
    }
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}

// ---- GLV pre-pass (BLS12-381 G1 only, as the reference: build_glv.js:3) ----------------------------
int msm_amd_glv_decompose_scalar(const uint8_t* scalar32, uint8_t* out64, uint32_t* sign) {
    if (!scalar32 || !out64 || !sign) return MSM_AMD_EINVAL;
    uint32_t k[8], k1[4], k2[4];
    memcpy(k, scalar32, 32);
    *sign = glv::decompose(k, k1, k2);
    memset(out64, 0, 64);
    memcpy(out64, k1, 16);
    memcpy(out64 + 32, k2, 16);
    return 0;
}
int msm_amd_glv_split(const uint8_t* scalar32, uint8_t* out32, uint32_t* sign) {
    if (!scalar32 || !out32 || !sign) return MSM_AMD_EINVAL;
    uint32_t k[8], k1[4], k2[4];
    memcpy(k, scalar32, 32);
    *sign = glv::split_centered(k, k1, k2);
    memcpy(out32, k1, 16);
    memcpy(out32 + 16, k2, 16);
    return 0;
}
int msm_amd_glv_endomorphism(const uint8_t* point_affine, uint32_t sign, uint8_t* out_affine) {
    if (!point_affine || !out_affine) return MSM_AMD_EINVAL;
    using HF = FpH<Bls12381Fp>;
    HF x, y, beta;
    memcpy(&x, point_affine, sizeof x);
    memcpy(&y, point_affine + sizeof x, sizeof y);
    memcpy(&beta, glv::Consts::BETA, sizeof beta);
    HF bx = HF::mul(x, beta);
    HF yy = sign ? y : HF::neg(y);
    memcpy(out_affine, &bx, sizeof bx);
    memcpy(out_affine + sizeof bx, &yy, sizeof yy);
    return 0;
}
int msm_amd_glv_preprocess(msm_amd_ctx* ctx, uint32_t flags, const void* points, const void* scalars, uint32_t n, void* points_out,
                           void* scalars_out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!points || !scalars || !points_out || !scalars_out) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t pb = (size_t)n * 96, sb = (size_t)n * 32;
    if (flags & MSM_AMD_DEVICE_PTRS) {
        if (((uintptr_t)points & 15) || ((uintptr_t)points_out & 15) || ((uintptr_t)scalars & 3) || ((uintptr_t)scalars_out & 3)) {
            msm_set_err(ctx, "device pointers must be 16-byte (points) / 4-byte (scalars) aligned");
            return MSM_AMD_EINVAL;
        }
        k_glv_preprocess<<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)points, (const uint32_t*)scalars, n, (uint8_t*)points_out,
                                                          (uint32_t*)scalars_out);
        HIPCHK(ctx, hipStreamSynchronize(st));
        HIPCHK(ctx, hipGetLastError());
        return 0;
    }
    void *dp = nullptr, *ds = nullptr, *dpo = nullptr, *dso = nullptr;
    HIPCHK(ctx, hipMalloc(&dp, pb));
    HIPCHK(ctx, hipMalloc(&ds, sb));
    HIPCHK(ctx, hipMalloc(&dpo, 2 * pb));
    HIPCHK(ctx, hipMalloc(&dso, 2 * sb));
    hipError_t e = hipMemcpyAsync(dp, points, pb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ds, scalars, sb, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_glv_preprocess<<<(n + 255) / 256, 256, 0, st>>>((const uint8_t*)dp, (const uint32_t*)ds, n, (uint8_t*)dpo, (uint32_t*)dso);
        e = hipMemcpyAsync(points_out, dpo, 2 * pb, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(scalars_out, dso, 2 * sb, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dp);
    (void)hipFree(ds);
    (void)hipFree(dpo);
    (void)hipFree(dso);
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("glv_preprocess: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    return 0;
}

int msm_amd_gen_points(msm_amd_ctx* ctx, int curve, int group, uint64_t seed, uint32_t n, void* dev_out) {
    return msm_amd_gen_points_range(ctx, curve, group, seed, 0, n, dev_out);
}
int msm_amd_gen_points_range(msm_amd_ctx* ctx, int curve, int group, uint64_t seed, uint32_t first, uint32_t n, void* dev_out) {
    if (!ctx || !dev_out) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (curve == MSM_AMD_BLS12381 && group == 1) {
        using P = Bls12381Fp;
        Affine<Fp<P>> g{fp_from<P>(P::G1X), fp_from<P>(P::G1Y)};
        return gen_points_t<Fp<P>>(ctx, g, seed, first, n, dev_out);
    }
    if (curve == MSM_AMD_BN128 && group == 1) {
        using P = Bn254Fp;
        Affine<Fp<P>> g{fp_from<P>(P::G1X), fp_from<P>(P::G1Y)};
        return gen_points_t<Fp<P>>(ctx, g, seed, first, n, dev_out);
    }
    if (curve == MSM_AMD_BLS12381 && group == 2) {
        using P = Bls12381Fp;
        Affine<Fp2<P>> g{{fp_from<P>(P::G2X0), fp_from<P>(P::G2X1)}, {fp_from<P>(P::G2Y0), fp_from<P>(P::G2Y1)}};
        return gen_points_t<Fp2<P>>(ctx, g, seed, first, n, dev_out);
    }
    if (curve == MSM_AMD_BN128 && group == 2) {
        using P = Bn254Fp;
        Affine<Fp2<P>> g{{fp_from<P>(P::G2X0), fp_from<P>(P::G2X1)}, {fp_from<P>(P::G2Y0), fp_from<P>(P::G2Y1)}};
        return gen_points_t<Fp2<P>>(ctx, g, seed, first, n, dev_out);
    }
    msm_set_err(ctx, "unknown curve/group");
    return MSM_AMD_EINVAL;
}

}  // extern "C"
