// prepared.hip -- prepared bases (round 3; include/msm_amd.h "prepared bases"): the window table of a fixed set of points, built once
// on the GPU.  The calls on a handle (msm_amd_multiexp_prepared[_async], engine.hip) are ordinary submissions of the engine in merged
// mode (Geom::merged): every window's digit into ONE bucket set, entry (i, w) addressing table point w * stride + i.
#include "engine_ctx.hpp"

// ------------------------------------------------------------------------------------------
// Prepared bases: the window table of a fixed set of points (the SRS of a prover; reference caller shape
// wasmcurves/benchmarks/multiexp.js:18-41: the same bases, call after call).
// table[w][i] = 2^(start bit of window w) * P_i, affine.  With it the digit of EVERY window of scalar i selects a point that goes into
// the SAME bucket set: the buckets, their reduction and the host's Horner loop shrink from W windows' worth to one, which in turn
// lets the windows be wider -- n = 2^20: 13 windows of 20 / 19 bits over 2^19 buckets instead of 16 windows x 2^15 buckets:
// 13 instead of 16 bucket additions per pair at the same number of buckets.
// ------------------------------------------------------------------------------------------
namespace msm {
// next[i] = 2^ndbl * prev[i]: K points per lane, ndbl doublings each in extended Jacobian coordinates, back to affine with one shared
// inversion per lane (Montgomery's trick, as k_jacobian_to_affine).  One-off work: loops are kept rolled.
template <class F>
__global__ void __launch_bounds__(256) k_table_next(const uint8_t* __restrict__ prev, uint32_t n, uint32_t ndbl, uint8_t* __restrict__ next) {
    constexpr int K = 4;
    const uint32_t first = (blockIdx.x * blockDim.x + threadIdx.x) * K;
    if (first >= n) return;
    XYZZ<F> p[K];
    F pref[K];
    F acc = F::one();
#pragma unroll 1
    for (int k = 0; k < K; ++k) {
        const uint32_t i = first + k;
        XYZZ<F> q = XYZZ<F>::inf();
        if (i < n) q = XYZZ<F>::from_affine(load16<Affine<F>>(prev + (size_t)i * sizeof(Affine<F>)));
#pragma unroll 1
        for (uint32_t d = 0; d < ndbl; ++d) q = XYZZ<F>::dbl(q);
        p[k] = q;
        pref[k] = acc;
        if (!q.is_inf()) acc = F::mul(acc, F::mul(q.zz, q.zzz));
    }
    F inv = F::inv(acc);
#pragma unroll 1
    for (int k = K - 1; k >= 0; --k) {
        const uint32_t i = first + k;
        if (i >= n) continue;
        Affine<F> a{F::zero(), F::zero()};
        if (!p[k].is_inf()) {
            const F t = F::mul(inv, pref[k]);          // 1 / (zz * zzz) of this point
            inv = F::mul(inv, F::mul(p[k].zz, p[k].zzz));
            a.x = F::mul(p[k].x, F::mul(t, p[k].zzz));   // x = X / ZZ
            a.y = F::mul(p[k].y, F::mul(t, p[k].zz));    // y = Y / ZZZ
        }
        store16(next + (size_t)i * sizeof(Affine<F>), Affine<F>::canon(a));
    }
}
}  // namespace msm

// Window layout of a table for n points and scalars below 2^bits: `ns` signed windows of c bits followed by W - ns unsigned ones of
// c - 1 bits (both kinds fill 2^(c-1) buckets evenly), the top one unsigned, W the fewest that cover `bits`, ns the fewest that do
// it with W.  The window size, measured (profiles/r03_sweep_prepared_window_bits.txt, BLS12-381 G1, ms per MSM with two calls in
// flight, c | ms: n = 2^16  14 .636, 15 .617, 16 .602, 17 .648;  2^18  16 1.167, 17 .983, 18 .981, 19 1.099;  2^20  18 3.05, 19 3.01,
// 20 2.55, 21 3.32;  2^22  19 11.4, 20 9.8, 21 10.5;  2^24  20 36.1, 21 38.7): c = log2 n up to 2^18, 20 from 2^19 on.  What decides is
// not only W x n bucket additions against 2^(c-1) buckets to reduce: 2^18 buckets are 4096 accumulation waves, 1.33 rounds of the
// 3072 that fit the chip, and the last third of a round costs almost a whole one (c = 19 at n = 2^20: 28 % more time per addition
// than c = 20, whose 8192 waves are 2.67 rounds -- the same as the 16 x 2^15 buckets of the general path); 2^20 buckets (c = 21) hold
// 13 entries each and the per-bucket costs show.  Below 2^16 the plain count decides: W x n additions + ~12 additions' worth per bucket.
static void plan_prepared(uint32_t n, uint32_t bits, uint32_t forced_c, uint32_t* c_out, uint32_t* W_out, uint32_t* ns_out) {
    auto layout = [&](uint32_t c, uint32_t* W, uint32_t* ns) {
        // W windows, ns of them signed (c bits), cover W (c - 1) + ns bits; ns <= W - 1 because the top window is unsigned
        *W = (bits + c) / c;                                     // = ceil((bits + 1) / c)
        *ns = bits > *W * (c - 1) ? bits - *W * (c - 1) : 0u;
        return (uint64_t)*W * (((uint64_t)n + 8191u) & ~8191ull) < (1ull << 31);   // (table rows are padded to 8192 points)
    };
    *c_out = *W_out = *ns_out = 0;
    uint32_t want = forced_c;
    if (!want && n >= (1u << 16)) {
        uint32_t lg = 0;
        while ((2ull << lg) <= n) ++lg;   // floor(log2 n)
        want = n >= (1u << 19) ? 20u : lg;
    }
    if (want && !forced_c && want > 4) {
        // the windows should tile the scalar width EXACTLY (W (c - 1) <= bits: then ns >= 0 signed windows make up the difference); a
        // width the wanted size overshoots -- 128-bit challenges at c = 20: 7 x 19 = 133 bits, the top window 14 of 19 bits wide, its
        // 2^14 buckets 32 times as dense as the rest -- takes the nearest size that does: 19 (7 windows, 2 of them signed)
        for (uint32_t c : {want, want - 1, want + 1, want - 2}) {
            uint32_t W, ns;
            if (c >= 3 && c <= (uint32_t)MAX_WINDOW_BITS && layout(c, &W, &ns) && (uint64_t)W * (c - 1) <= bits) {
                want = c;
                break;
            }
        }
    }
    if (want) {
        // (a table too large for 31-bit entry indices at the wanted size takes the nearest wider window that fits)
        for (uint32_t c = std::min<uint32_t>(want, MAX_WINDOW_BITS); c <= (uint32_t)MAX_WINDOW_BITS; ++c)
            if (layout(c, W_out, ns_out)) {
                *c_out = c;
                return;
            }
        *W_out = *ns_out = 0;
        return;
    }
    uint64_t best = ~0ull;
    for (uint32_t c = 3; c <= 16u; ++c) {
        uint32_t W, ns;
        if (!layout(c, &W, &ns)) continue;
        const uint64_t cost = (uint64_t)W * n + 12ull * (1ull << (c - 1));
        if (cost < best) {
            best = cost;
            *c_out = c;
            *W_out = W;
            *ns_out = ns;
        }
    }
}

template <class F>
static int build_table(msm_amd_ctx* ctx, msm_amd_bases* h, const void* bases, uint32_t flags) {
    const size_t ASZ = sizeof(Affine<F>), JSZ = sizeof(Jacobian<F>);
    const uint32_t n = h->n;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    h->stride = (n + 8191u) & ~8191u;   // the sort's largest tile (8192 scalars, round 4): a tile never straddles an index slice
    const size_t row = (size_t)h->stride * ASZ;
    h->table_bytes = (size_t)h->W * row;
    HIPCHK(ctx, hipMalloc(&h->table, h->table_bytes));
    // (after the table has taken its share of the device: the doubling kernel's scratch must still fit -- engine_ctx.hpp)
    if (int rc = msm_guard(ctx, 1, &k_table_next<F>, "prepare_bases")) return rc;
    hipStream_t st = ctx->stream_s;
    uint8_t* t0 = (uint8_t*)h->table;
    const bool dev = (flags & (MSM_AMD_DEVICE_PTRS | MSM_AMD_DEVICE_BASES)) != 0;
    if (flags & MSM_AMD_BASES_JACOBIAN) {
        void* dj = const_cast<void*>(bases);
        if (!dev) {
            HIPCHK(ctx, hipMalloc(&dj, (size_t)n * JSZ));
            if (hipMemcpyAsync(dj, bases, (size_t)n * JSZ, hipMemcpyHostToDevice, st) != hipSuccess) {
                (void)hipFree(dj);
                msm_set_err(ctx, "prepare_bases: copy of the bases failed");
                return MSM_AMD_EHIP;
            }
        }
        k_jacobian_to_affine<F><<<((n + jacobian_to_affine_points_per_lane<F>() - 1) / jacobian_to_affine_points_per_lane<F>() + 255) / 256, 256, 0, st>>>((const uint8_t*)dj, n, t0);
        hipError_t e = hipStreamSynchronize(st);
        if (!dev) (void)hipFree(dj);
        HIPCHK(ctx, e);
    } else {
        HIPCHK(ctx, hipMemcpyAsync(t0, bases, (size_t)n * ASZ, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    Geom g{};
    g.c = h->c;
    g.ns = h->ns;
    for (uint32_t w = 1; w < h->W; ++w) {
        const uint32_t ndbl = window_start_bit(g, w) - window_start_bit(g, w - 1);
        k_table_next<F><<<((n + 3) / 4 + 255) / 256, 256, 0, st>>>(t0 + (size_t)(w - 1) * row, n, ndbl, t0 + (size_t)w * row);
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

extern "C" {

// ---- prepared bases --------------------------------------------------------------------------------------
int msm_amd_prepare_bases(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, uint32_t n, uint32_t scalar_bits,
                          msm_amd_bases** out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    *out = nullptr;
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || !bases || n == 0 || scalar_bits > 512) {
        msm_set_err(ctx, "prepare_bases: bad curve / group / bases / n / scalar_bits");
        return MSM_AMD_EINVAL;
    }
    if ((flags & (MSM_AMD_DEVICE_PTRS | MSM_AMD_DEVICE_BASES)) && ((uintptr_t)bases & 15)) {
        msm_set_err(ctx, "device bases pointer must be 16-byte aligned");
        return MSM_AMD_EINVAL;
    }
    // 0: scalars reduced modulo the group order, what a prover's witness and quotient coefficients are (255 bits for BLS12-381, 254 for BN254)
    if (scalar_bits == 0) scalar_bits = curve == MSM_AMD_BLS12381 ? 255 : 254;
    msm_amd_bases* h = new (std::nothrow) msm_amd_bases();
    if (!h) return MSM_AMD_ENOMEM;
    h->ctx = ctx;
    h->curve = curve;
    h->group = group;
    h->n = n;
    h->scalar_bits = scalar_bits;
    plan_prepared(n, scalar_bits, ctx->force_c >= 3 ? ctx->force_c : 0, &h->c, &h->W, &h->ns);
    if (h->c == 0) {
        delete h;
        msm_set_err(ctx, "prepare_bases: windows x points of the table must stay below 2^31");
        return MSM_AMD_EINVAL;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);   // (uses stream S; no call may be submitted meanwhile)
    int rc;
    if (curve == MSM_AMD_BLS12381 && group == 1) rc = build_table<Fp<Bls12381Fp>>(ctx, h, bases, flags);
    else if (curve == MSM_AMD_BLS12381) rc = build_table<Fp2<Bls12381Fp>>(ctx, h, bases, flags);
    else if (group == 1) rc = build_table<Fp<Bn254Fp>>(ctx, h, bases, flags);
    else rc = build_table<Fp2<Bn254Fp>>(ctx, h, bases, flags);
    if (rc) {
        if (h->table) (void)hipFree(h->table);
        delete h;
        return rc;
    }
    ctx->prepared_handles.push_back(h);
    *out = h;
    return 0;
}

int msm_amd_release_bases(msm_amd_ctx* ctx, msm_amd_bases* h) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (!h) return 0;
    {
        // One critical section with submit() (engine.hip), which counts a call in under the same lock after checking that its table is
        // still registered: a table is either released or gets a new call, never both (a thread sharing the table could otherwise
        // submit between the in-flight check and the free, and its kernels would gather from freed memory).
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto& v = ctx->prepared_handles;
        if (std::find(v.begin(), v.end(), h) == v.end()) {   // (also: a handle of another context, or one released before -- never dereferenced)
            msm_set_err(ctx, "release_bases: not a handle of this context (another context's, or already released)");
            return MSM_AMD_EINVAL;
        }
        if (h->in_flight.load() > 0) {
            msm_set_err(ctx, "release_bases: calls on these bases are still in flight (wait for their tickets first)");
            return MSM_AMD_EINVAL;
        }
        v.erase(std::remove(v.begin(), v.end(), h), v.end());
    }
    (void)hipSetDevice(ctx->device);
    if (h->table) (void)hipFree(h->table);
    delete h;
    return 0;
}

// The layout msm_amd_prepare_bases would choose, and what the table would take -- without building it (no GPU, no context).
int msm_amd_plan_prepared(int curve, int group, uint32_t n, uint32_t scalar_bits, uint32_t window_bits, uint32_t* out6, uint64_t* table_bytes) {
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || n == 0 || scalar_bits > 512 || !out6) return MSM_AMD_EINVAL;
    if (scalar_bits == 0) scalar_bits = curve == MSM_AMD_BLS12381 ? 255 : 254;
    uint32_t c = 0, W = 0, ns = 0;
    plan_prepared(n, scalar_bits, window_bits >= 3 ? window_bits : 0, &c, &W, &ns);
    if (c == 0) return MSM_AMD_EINVAL;   // windows x points of the table would not stay below 2^31
    out6[0] = n;
    out6[1] = scalar_bits;
    out6[2] = c;
    out6[3] = W;
    out6[4] = ns;
    out6[5] = 1u << (c - 1);
    const uint64_t affine = (uint64_t)2 * (curve == MSM_AMD_BLS12381 ? 48 : 32) * (uint64_t)group;
    if (table_bytes) *table_bytes = (uint64_t)W * (((uint64_t)n + 8191u) & ~8191ull) * affine;
    return 0;
}

int msm_amd_bases_info(const msm_amd_bases* h, uint32_t* out6, uint64_t* table_bytes) {
    if (!h || !out6) return MSM_AMD_EINVAL;
    out6[0] = h->n;
    out6[1] = h->scalar_bits;
    out6[2] = h->c;
    out6[3] = h->W;
    out6[4] = h->ns;
    out6[5] = 1u << (h->c - 1);
    if (table_bytes) *table_bytes = h->table_bytes;
    return 0;
}

}  // extern "C"
