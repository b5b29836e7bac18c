// msm_engine.hip -- MI355X (gfx950) Pippenger multi-scalar multiplication behind the C ABI of
// include/msm_amd.h.
//
// What the reference does (wasmcurves/src/build_multiexp.js): for each c-bit window, walk the n
// (scalar, base) pairs sequentially, add each base into table[digit-1] with g1m_addMixed
// (:186-233), collapse the table by recursive halving (:373-461), and Horner-combine the windows
// (:310-369).  One thread, unsigned digits, table in WASM linear memory.
//
// What this engine does instead (same group element, different schedule):
//   1. digits      one thread per scalar: signed radix-2^c recoding of the requested bit range
//                  (digit in [-2^(c-1), 2^(c-1)]), histogram of bucket populations
//   2. scan/order  exclusive scan of the histogram; buckets ordered by population so that the
//                  64 lanes of a wavefront run the same trip count
//   3. scatter     radix-c bucket scatter: point indices (sign in bit 31) grouped by bucket
//   4. accumulate  one lane per bucket: gather affine points, XYZZ mixed additions (curve.hpp)
//   5. reduce      per window sum_k k*B_k: short running sums per lane, a small scalar
//                  multiplication by the lane's bucket offset, LDS tree reduction per workgroup
//   6. combine     window partials (a few KB) go to the host, which Horner-combines them and
//                  normalises the result (the serial doubling chain is latency-, not throughput-bound)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/msm_amd.h"
#include "engine_common.hpp"
#include "field_host.hpp"
#include "glv.hpp"

namespace msm {


// launchers of the bucket kernels, instantiated in bucket_g1_bls12381.hip / bucket_g1_bn254.hip / bucket_g2.hip
template <class F, int VARIANT>
void launch_accumulate(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                       const uint32_t* order, uint32_t nb, uint8_t* buckets, uint32_t seg, void* ctl, void* work, void* biglist);
template <class F>
void launch_accumulate_split2(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                              const uint32_t* order, uint32_t nb, uint8_t* buckets, uint32_t seg, void* ctl, void* work, void* biglist);
template <class F>
void launch_accumulate_tail(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset,
                            const uint32_t* count, uint8_t* buckets, uint32_t seg, void* ctl, void* work, void* biglist, uint8_t* seg_out,
                            uint32_t tail_block);
template <class F>
void launch_bucket_reduce(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                          uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials, const uint32_t* offset, const uint32_t* count,
                          uint32_t gb0, uint32_t NB);
// field flavour used inside the bucket kernels: G1 fields expand their multiplies inline
template <class F>
size_t bucket_bytes();   // bytes of one bucket in HBM (the bucket kernels keep their own representation, curve_u.hpp)
template <class F>
uint32_t lanes_per_point();   // threads that share one point in the bucket kernels (2 for G2: Fp2 components on lane pairs)
template <class F>
struct KernelField {
    using type = F;
    static constexpr bool chained_variant = false;
};
template <class P>
struct KernelField<Fp<P, false>> {
    using type = Fp<P, true>;
    static constexpr bool chained_variant = true;   // bucket_g1_*_acc.hip
};

// Bits [pos, pos+nb) of scalar i (nb <= 25, as build_multiexp.js:25-94); bytes past the scalar read 0.
MSM_D uint32_t scalar_bits(const uint8_t* scalars, const Geom& g, uint32_t i, uint32_t pos, uint32_t nb) {
    if (nb == 0) return 0;
    const uint8_t* s = scalars + (size_t)i * g.scalar_size;
    uint32_t byte = pos >> 3;
    uint64_t w = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b)
        if (byte + b < g.scalar_size) w |= (uint64_t)s[byte + b] << (8 * b);
    return (uint32_t)(w >> (pos & 7)) & ((1u << nb) - 1u);
}
// fast path: scalar_size % 4 == 0 and 4-byte aligned scalars
MSM_D uint32_t scalar_bits32(const uint32_t* scalars32, const Geom& g, uint32_t i, uint32_t pos, uint32_t nb) {
    if (nb == 0) return 0;
    uint32_t words = g.scalar_size >> 2;
    const uint32_t* s = scalars32 + (size_t)i * words;
    uint32_t wi = pos >> 5;
    uint32_t w0 = s[wi];
    uint32_t w1 = (wi + 1 < words) ? s[wi + 1] : 0u;
    uint64_t w = ((uint64_t)w1 << 32) | w0;
    return (uint32_t)(w >> (pos & 31)) & ((1u << nb) - 1u);
}

// Signed radix-2^c recoding of scalar i; calls f(window, bucket_in_window (0-based), negative).
template <bool FAST, class Fn>
MSM_D void for_each_digit(const uint8_t* scalars, const Geom& g, uint32_t i, Fn&& f) {
    uint32_t carry = 0;
    for (uint32_t w = 0; w < g.W; ++w) {
        uint32_t off = w * g.c;
        uint32_t nb = off >= g.nbits ? 0u : min(g.c, g.nbits - off);
        uint32_t raw = FAST ? scalar_bits32(reinterpret_cast<const uint32_t*>(scalars), g, i, g.bit_lo + off, nb)
                            : scalar_bits(scalars, g, i, g.bit_lo + off, nb);
        raw += carry;
        bool negd = raw > g.half;
        uint32_t mag = negd ? (1u << g.c) - raw : raw;
        carry = negd ? 1u : 0u;
        if (mag != 0 && (w % g.nshards) == g.shard) f(w / g.nshards, mag - 1, negd);
    }
}

// ------------------------------------------------------------------------------------------
// stage 1 / 3: histogram and scatter
// ------------------------------------------------------------------------------------------
template <bool FAST>
__global__ void __launch_bounds__(256) k_histogram(const uint8_t* __restrict__ scalars, Geom g, uint32_t* __restrict__ count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    for_each_digit<FAST>(scalars, g, i, [&](uint32_t wl, uint32_t b, bool) { atomicAdd(&count[wl * g.half + b], 1u); });
}

template <bool FAST>
__global__ void __launch_bounds__(256) k_scatter(const uint8_t* __restrict__ scalars, Geom g, uint32_t* __restrict__ cursor,
                                                  uint32_t* __restrict__ sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    for_each_digit<FAST>(scalars, g, i, [&](uint32_t wl, uint32_t b, bool negd) {
        uint32_t pos = atomicAdd(&cursor[wl * g.half + b], 1u);
        sorted[pos] = i | (negd ? 0x80000000u : 0u);
    });
}


// ------------------------------------------------------------------------------------------
// stage 1-3, LDS path (default): digits -> two-pass radix partition by bucket.
//   k_digits          one lane per scalar, all windows: 16-bit signed-digit codes, window-major
//                     (code = sign<<15 | (magnitude-1); 0xFFFF = zero digit)
//   k_coarse_hist     per (tile, window): LDS histogram over the top bits of the bucket index
//   (scan)            exclusive scan of the (window, coarse bin, tile) count matrix
//   k_coarse_scatter  per (tile, window): local counting sort in LDS, then each (tile, bin) run is
//                     written with consecutive lanes -> coalesced 8-byte entries {fine key, index}
//   k_fine_sort       one workgroup per (window, coarse bin): histogram of the low bucket bits,
//                     bucket count/offset tables, scatter of the point indices inside the bin's
//                     (L2-resident) span of `sorted`
// No global atomics; every global access is a coalesced stream or a scatter confined to one bin.
// ------------------------------------------------------------------------------------------
constexpr int SORT_BLOCK = 256;
constexpr int SORT_ITEMS = 16;
constexpr int SORT_TILE = SORT_BLOCK * SORT_ITEMS;   // digits per workgroup
constexpr int COARSE_MAX = 128;                      // coarse bins per window (c <= 16, 8 fine bits)
constexpr uint32_t DIGIT_ZERO = 0xFFFFu;

template <bool FAST>
__global__ void __launch_bounds__(256) k_digits(const uint8_t* __restrict__ scalars, Geom g, uint16_t* __restrict__ dig) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < g.W; ++w) {
        uint32_t off = w * g.c;
        uint32_t nb = off >= g.nbits ? 0u : min(g.c, g.nbits - off);
        uint32_t raw = FAST ? scalar_bits32(reinterpret_cast<const uint32_t*>(scalars), g, i, g.bit_lo + off, nb)
                            : scalar_bits(scalars, g, i, g.bit_lo + off, nb);
        raw += carry;
        bool negd = raw > g.half;
        uint32_t mag = negd ? (1u << g.c) - raw : raw;
        carry = negd ? 1u : 0u;
        if ((w % g.nshards) == g.shard) {
            uint32_t code = mag == 0 ? DIGIT_ZERO : ((negd ? 0x8000u : 0u) | (mag - 1));
            dig[(size_t)(w / g.nshards) * g.n + i] = (uint16_t)code;
        }
    }
}

// c == 16, 32-byte aligned scalars, whole scalar: the raw digits are the 16 half-words of the scalar (two 16-byte loads)
__global__ void __launch_bounds__(256) k_digits_c16(const uint8_t* __restrict__ scalars, Geom g, uint16_t* __restrict__ dig) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars) + (size_t)i * 2;
    uint4 lo = sp[0], hi = sp[1];
    uint32_t wd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t carry = 0;
#pragma unroll
    for (uint32_t w = 0; w < 17; ++w) {
        uint32_t raw = w < 16 ? ((wd[w >> 1] >> (16 * (w & 1))) & 0xFFFFu) : 0u;
        raw += carry;
        bool negd = raw > 0x8000u;
        uint32_t mag = negd ? 0x10000u - raw : raw;
        carry = negd ? 1u : 0u;
        if ((w % g.nshards) == g.shard) {
            uint32_t code = mag == 0 ? DIGIT_ZERO : ((negd ? 0x8000u : 0u) | (mag - 1));
            dig[(size_t)(w / g.nshards) * g.n + i] = (uint16_t)code;
        }
    }
}

// GLV path (BLS12-381 G1, c = 16, whole 32-byte scalars; SURVEY.md 8f rank 1, constants of build_glv.js:15-22): every
// scalar is split into two signed 127-bit halves (glv::split_centered).  Virtual item 2i is (P_i, k1), item 2i+1 is
// (phi(P_i), k2) with phi(x, y) = (beta*x, y): 2n items, 8 windows and no carry out of the last one -- the same number of
// bucket additions as 16 windows over n items, but half the buckets to initialise, reduce and Horner-combine.
__global__ void __launch_bounds__(256) k_digits_glv(const uint8_t* __restrict__ scalars, uint32_t n, Geom g, uint16_t* __restrict__ dig) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars) + (size_t)i * 2;
    uint4 lo = sp[0], hi = sp[1];
    uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t h[2][4];
    uint32_t sg = glv::split_centered(k, h[0], h[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        uint32_t carry = 0;
        const bool sneg = ((sg >> j) & 1u) != 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; ++w) {
            uint32_t raw = (h[j][w >> 1] >> (16 * (w & 1))) & 0xFFFFu;
            raw += carry;
            // raw == 2^15 may be written +2^15 or -2^15 (+carry): take the form whose final sign is positive, because the
            // code "negative, magnitude 2^15" (0xFFFF) is the zero-digit marker.  The halves are below 2^127 (top digit
            // <= 0x5623), so this never carries out of the last window.
            bool negd = raw > 0x8000u || (raw == 0x8000u && sneg);
            uint32_t mag = negd ? 0x10000u - raw : raw;
            carry = negd ? 1u : 0u;
            if ((w % g.nshards) == g.shard) {
                uint32_t code = mag == 0 ? DIGIT_ZERO : (((negd != sneg) ? 0x8000u : 0u) | (mag - 1));
                dig[(size_t)(w / g.nshards) * g.n + 2 * i + j] = (uint16_t)code;
            }
        }
    }
}
// x-coordinates of the endomorphism images: xbeta[i] = beta * x_i (y is shared with the original point)
template <class F>
__global__ void __launch_bounds__(256) k_glv_xbeta(const uint8_t* __restrict__ affine, uint32_t n, uint8_t* __restrict__ xbeta) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = load16<F>(affine + (size_t)i * sizeof(Affine<F>));
    F beta;
#pragma unroll
    for (int j = 0; j < F::N; ++j) beta.v[j] = glv::Consts::BETA[j];
    store16(xbeta + (size_t)i * sizeof(F), F::canon(F::mul(x, beta)));
}
template <class F>
struct GlvField {
    static constexpr bool value = false;
};
template <>
struct GlvField<Fp<Bls12381Fp>> {
    static constexpr bool value = true;
};

__global__ void __launch_bounds__(SORT_BLOCK) k_coarse_hist(const uint16_t* __restrict__ dig, uint32_t n, uint32_t fine_bits, uint32_t cb,
                                                              uint32_t ntile, uint32_t* __restrict__ M) {
    __shared__ uint32_t h[COARSE_MAX];
    const uint32_t wl = blockIdx.y, tile = blockIdx.x;
    if (threadIdx.x < COARSE_MAX) h[threadIdx.x] = 0;
    __syncthreads();
    const uint16_t* d = dig + (size_t)wl * n;
    const uint32_t base = tile * SORT_TILE;
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        uint32_t i = base + k * SORT_BLOCK + threadIdx.x;
        if (i < n) {
            uint32_t code = d[i];
            if (code != DIGIT_ZERO) atomicAdd(&h[(code & 0x7FFFu) >> fine_bits], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < cb) M[((size_t)wl * cb + threadIdx.x) * ntile + tile] = h[threadIdx.x];
}

__global__ void __launch_bounds__(SORT_BLOCK) k_coarse_scatter(const uint16_t* __restrict__ dig, uint32_t n, uint32_t fine_bits, uint32_t cb,
                                                                 uint32_t ntile, const uint32_t* __restrict__ Mscan, uint64_t* __restrict__ tmp) {
    __shared__ uint32_t h[COARSE_MAX];        // counts, then running cursors
    __shared__ uint32_t lstart[COARSE_MAX];   // start of each bin inside the staged tile
    __shared__ uint32_t delta[COARSE_MAX];    // global position of slot s = delta[bin] + s
    __shared__ uint64_t stage[SORT_TILE];
    const uint32_t wl = blockIdx.y, tile = blockIdx.x;
    if (threadIdx.x < COARSE_MAX) h[threadIdx.x] = 0;
    __syncthreads();
    const uint16_t* d = dig + (size_t)wl * n;
    const uint32_t base = tile * SORT_TILE;
    uint32_t codes[SORT_ITEMS];
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        uint32_t i = base + k * SORT_BLOCK + threadIdx.x;
        codes[k] = i < n ? (uint32_t)d[i] : DIGIT_ZERO;
        if (codes[k] != DIGIT_ZERO) atomicAdd(&h[(codes[k] & 0x7FFFu) >> fine_bits], 1u);
    }
    __syncthreads();
    // exclusive scan of the (<= 128) bin counts by the first two waves
    if (threadIdx.x < COARSE_MAX) lstart[threadIdx.x] = threadIdx.x < cb ? h[threadIdx.x] : 0;
    __syncthreads();
    for (int dd = 1; dd < COARSE_MAX; dd <<= 1) {
        uint32_t t = 0;
        if (threadIdx.x < COARSE_MAX && (int)threadIdx.x >= dd) t = lstart[threadIdx.x - dd];
        __syncthreads();
        if (threadIdx.x < COARSE_MAX) lstart[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t incl = 0, cntb = 0;
    if (threadIdx.x < COARSE_MAX) {
        incl = lstart[threadIdx.x];
        cntb = threadIdx.x < cb ? h[threadIdx.x] : 0;
    }
    __syncthreads();
    if (threadIdx.x < COARSE_MAX) {
        uint32_t excl = incl - cntb;
        lstart[threadIdx.x] = excl;
        if (threadIdx.x < cb) delta[threadIdx.x] = Mscan[((size_t)wl * cb + threadIdx.x) * ntile + tile] - excl;   // mod 2^32
        h[threadIdx.x] = 0;
    }
    __syncthreads();
    const uint32_t total = lstart[cb - 1];   // start of the last bin; its count is added below
#pragma unroll
    for (int k = 0; k < SORT_ITEMS; ++k) {
        if (codes[k] != DIGIT_ZERO) {
            uint32_t i = base + k * SORT_BLOCK + threadIdx.x;
            uint32_t b = codes[k] & 0x7FFFu;
            uint32_t bin = b >> fine_bits;
            uint32_t fine = b & ((1u << fine_bits) - 1u);
            uint32_t r = atomicAdd(&h[bin], 1u);
            uint64_t e = ((uint64_t)bin << 40) | ((uint64_t)fine << 32) | (uint64_t)(i | ((codes[k] & 0x8000u) << 16));
            stage[lstart[bin] + r] = e;
        }
    }
    __syncthreads();
    const uint32_t staged = total + h[cb - 1];
    for (uint32_t s = threadIdx.x; s < staged; s += SORT_BLOCK) {
        uint64_t e = stage[s];
        uint32_t bin = (uint32_t)(e >> 40);
        tmp[(uint32_t)(delta[bin] + s)] = e;
    }
}

__global__ void __launch_bounds__(SORT_BLOCK) k_fine_sort(const uint64_t* __restrict__ tmp, const uint32_t* __restrict__ Mscan, uint32_t ntile,
                                                            uint32_t nbins_total, uint32_t fine_bits, uint32_t cb, uint32_t half,
                                                            uint32_t* __restrict__ sorted, uint32_t* __restrict__ count, uint32_t* __restrict__ offset) {
    // One workgroup per (window, coarse bin).  Sweep 1: histogram of the fine key -> bucket count/offset.
    // Sweep 2: the bin is walked in tiles of SORT_TILE entries; each tile is counting-sorted in LDS and
    // every (tile, bucket) run is then stored by consecutive lanes, so the writes into `sorted` are
    // coalesced runs instead of 4-byte scatters (which cost 6x write amplification, profiles/r01_pmc_*).
    __shared__ uint32_t h[256];        // bucket populations of the whole bin
    __shared__ uint32_t lofs[256];     // bucket start inside the bin, advanced tile by tile
    __shared__ uint32_t th[256];       // tile: counts, then cursors
    __shared__ uint32_t tstart[256];   // tile: start of each bucket inside the staged tile
    __shared__ uint32_t delta[256];    // global position of staged slot s = delta[bucket] + s
    __shared__ uint64_t stage[SORT_TILE];
    const uint32_t bin = blockIdx.x;                          // = wl * cb + coarse bin
    const uint32_t start = Mscan[(size_t)bin * ntile];
    const uint32_t end = Mscan[(size_t)(bin + 1) * ntile];    // the scan carries one extra element = total
    const uint32_t nf = 1u << fine_bits;
    const uint32_t t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    for (uint32_t p = start + t; p < end; p += SORT_BLOCK) atomicAdd(&h[(uint32_t)(tmp[p] >> 32) & 0xFFu], 1u);
    __syncthreads();
    uint32_t v = h[t];
    lofs[t] = v;
    __syncthreads();
    for (int dd = 1; dd < 256; dd <<= 1) {
        uint32_t x = (int)t >= dd ? lofs[t - dd] : 0;
        __syncthreads();
        lofs[t] += x;
        __syncthreads();
    }
    uint32_t excl = lofs[t] - v;
    __syncthreads();
    lofs[t] = excl;
    if (t < nf) {
        uint32_t wl = bin / cb, cbin = bin % cb;
        uint32_t gb = wl * half + (cbin << fine_bits) + t;
        count[gb] = v;
        offset[gb] = start + excl;
    }
    __syncthreads();
    for (uint32_t tb = start; tb < end; tb += SORT_TILE) {
        const uint32_t tn = min((uint32_t)SORT_TILE, end - tb);
        th[t] = 0;
        __syncthreads();
        uint64_t e[SORT_ITEMS];
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; ++k) {
            uint32_t s = k * SORT_BLOCK + t;
            e[k] = s < tn ? tmp[tb + s] : ~0ull;
            if (s < tn) atomicAdd(&th[(uint32_t)(e[k] >> 32) & 0xFFu], 1u);
        }
        __syncthreads();
        uint32_t c = th[t];
        tstart[t] = c;
        __syncthreads();
        for (int dd = 1; dd < 256; dd <<= 1) {
            uint32_t x = (int)t >= dd ? tstart[t - dd] : 0;
            __syncthreads();
            tstart[t] += x;
            __syncthreads();
        }
        uint32_t ts = tstart[t] - c;
        __syncthreads();
        tstart[t] = ts;
        delta[t] = start + lofs[t] - ts;   // mod 2^32
        lofs[t] += c;                      // next tile of this bucket continues here
        th[t] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SORT_ITEMS; ++k) {
            uint32_t s = k * SORT_BLOCK + t;
            if (s < tn) {
                uint32_t f = (uint32_t)(e[k] >> 32) & 0xFFu;
                uint32_t r = atomicAdd(&th[f], 1u);
                stage[tstart[f] + r] = e[k];
            }
        }
        __syncthreads();
        for (uint32_t s = t; s < tn; s += SORT_BLOCK) {
            uint64_t x = stage[s];
            uint32_t f = (uint32_t)(x >> 32) & 0xFFu;
            sorted[(uint32_t)(delta[f] + s)] = (uint32_t)x;
        }
        __syncthreads();
    }
    (void)nbins_total;
}

// ------------------------------------------------------------------------------------------
// stage 2: exclusive scan of bucket populations (three small kernels) and ordering by population
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 16;                       // per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;   // 4096 per workgroup

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile_sums(const uint32_t* __restrict__ count, uint32_t nb, uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t red[SCAN_BLOCK];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < nb) s += count[base + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int d = SCAN_BLOCK / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = red[0];
}
// single workgroup: in-place exclusive scan of the tile sums
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(uint32_t* __restrict__ tile_sums, uint32_t ntiles) {
    __shared__ uint32_t buf[SCAN_BLOCK];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < ntiles; start += SCAN_BLOCK) {
        uint32_t idx = start + threadIdx.x;
        uint32_t v = idx < ntiles ? tile_sums[idx] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
            uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = buf[threadIdx.x];
        if (idx < ntiles) tile_sums[idx] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry += incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* __restrict__ count, uint32_t nb, const uint32_t* __restrict__ tile_sums,
                                                            uint32_t* __restrict__ offset, uint32_t* __restrict__ cursor) {
    __shared__ uint32_t buf[SCAN_BLOCK];
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t loc[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        loc[k] = (base + k < nb) ? count[base + k] : 0;
        s += loc[k];
    }
    buf[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
        uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = tile_sums[blockIdx.x] + buf[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < nb) {
            offset[base + k] = run;
            cursor[base + k] = run;
        }
        run += loc[k];
    }
}

// Ordering of buckets by population (descending), counting sort on min(count, ORDER_BINS-1).
constexpr int ORDER_BINS = 1024;
// Sort key: (bucket >= split ? 1 : 0, size class).  With split < nb the order array comes out as
// [buckets of the first window group by falling population][buckets of the second group by falling population],
// which lets the engine accumulate and reduce the two groups in two overlapping waves (see submit_msm).
constexpr int ORDER_KEYS = 2 * ORDER_BINS;
MSM_D uint32_t order_key(uint32_t b, uint32_t cnt, uint32_t split) {
    return (b >= split ? (uint32_t)ORDER_BINS : 0u) + ((ORDER_BINS - 1) - min(cnt, (uint32_t)(ORDER_BINS - 1)));   // big buckets first
}
__global__ void __launch_bounds__(1024) k_order_hist(const uint32_t* __restrict__ count, uint32_t nb, uint32_t split, uint32_t* __restrict__ bins) {
    __shared__ uint32_t h[ORDER_KEYS];
    h[threadIdx.x] = 0;
    h[threadIdx.x + ORDER_BINS] = 0;
    __syncthreads();
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) atomicAdd(&h[order_key(b, count[b], split)], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&bins[threadIdx.x], h[threadIdx.x]);
    if (h[threadIdx.x + ORDER_BINS]) atomicAdd(&bins[threadIdx.x + ORDER_BINS], h[threadIdx.x + ORDER_BINS]);
}
__global__ void __launch_bounds__(1024) k_order_scan(uint32_t* __restrict__ bins) {
    __shared__ uint32_t buf[ORDER_BINS];
    // two keys per lane (2t, 2t+1), then a scan over the pair sums
    uint32_t a = bins[2 * threadIdx.x], b = bins[2 * threadIdx.x + 1];
    uint32_t v = a + b;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < ORDER_BINS; d <<= 1) {
        uint32_t t = (int)threadIdx.x >= d ? buf[threadIdx.x - d] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t excl = buf[threadIdx.x] - v;
    bins[2 * threadIdx.x] = excl;
    bins[2 * threadIdx.x + 1] = excl + a;
}
__global__ void __launch_bounds__(1024) k_order_scatter(const uint32_t* __restrict__ count, uint32_t nb, uint32_t split, uint32_t* __restrict__ bins,
                                                         uint32_t* __restrict__ order) {
    __shared__ uint32_t h[ORDER_KEYS];
    __shared__ uint32_t basep[ORDER_KEYS];
    h[threadIdx.x] = 0;
    h[threadIdx.x + ORDER_BINS] = 0;
    __syncthreads();
    // each workgroup owns a contiguous span of buckets
    uint32_t per = (nb + gridDim.x - 1) / gridDim.x;
    uint32_t lo = blockIdx.x * per, hi = min(nb, lo + per);
    for (uint32_t b = lo + threadIdx.x; b < hi; b += blockDim.x) atomicAdd(&h[order_key(b, count[b], split)], 1u);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < ORDER_KEYS; k += ORDER_BINS) basep[k] = h[k] ? atomicAdd(&bins[k], h[k]) : 0;
    __syncthreads();
    h[threadIdx.x] = 0;
    h[threadIdx.x + ORDER_BINS] = 0;
    __syncthreads();
    for (uint32_t b = lo + threadIdx.x; b < hi; b += blockDim.x) {
        uint32_t key = order_key(b, count[b], split);
        uint32_t p = basep[key] + atomicAdd(&h[key], 1u);
        order[p] = b;
    }
}

// ------------------------------------------------------------------------------------------
// Jacobian bases -> affine (g?m_multiexp entry point): Montgomery's trick, 8 points per lane
// (build_curve_jacobian_a0.js:1040-1125 batchToAffine does the same thing sequentially).
// ------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_jacobian_to_affine(const uint8_t* __restrict__ jac, uint32_t n, uint8_t* __restrict__ aff) {
    constexpr int K = 8;
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t first = t * K;
    if (first >= n) return;
    F pref[K];
    F acc = F::one();
    for (int k = 0; k < K; ++k) {
        uint32_t i = first + k;
        pref[k] = acc;
        if (i < n) {
            F z = load16<F>(jac + (size_t)i * 3 * sizeof(F) + 2 * sizeof(F));
            if (!z.is_zero()) acc = F::mul(acc, z);
        }
    }
    F inv = F::inv(acc);
    for (int k = K - 1; k >= 0; --k) {
        uint32_t i = first + k;
        if (i >= n) continue;
        Jacobian<F> p = load16<Jacobian<F>>(jac + (size_t)i * sizeof(Jacobian<F>));
        Affine<F> a;
        if (p.z.is_zero()) {
            a = {F::zero(), F::zero()};
        } else {
            F zi = F::mul(inv, pref[k]);
            inv = F::mul(inv, p.z);
            F zi2 = F::sqr(zi);
            a.x = F::mul(p.x, zi2);
            a.y = F::mul(p.y, F::mul(zi2, zi));
        }
        store16(aff + (size_t)i * sizeof(Affine<F>), Affine<F>::canon(a));
    }
}



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
__global__ void __launch_bounds__(256) k_gen_points(Affine<F> gen, uint64_t seed, uint32_t n, uint8_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t k[4];
    for (int j = 0; j < 4; ++j) k[j] = splitmix64(seed * 0x100000001B3ull + (uint64_t)i * 4 + j);
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

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace msm

using namespace msm;

// One in-flight MSM: its scratch buffers, events and what the host needs to finish it.  A context owns
// two slots so that the bucket reduction of one call (latency-bound, one wave per SIMD, stream B) overlaps
// the sort + accumulation of the next call (throughput-bound, stream A).
struct Slot {
    DevBuf d_dig, d_tmp, d_M, d_Mscan, d_tailctl, d_work, d_biglist, d_segout;
    DevBuf d_xbeta;   // GLV path: beta * x of every base
    DevBuf d_bases, d_scalars, d_affine, d_count, d_offset, d_cursor, d_order, d_bins, d_tiles, d_sorted, d_buckets, d_partials;
    void* h_partials = nullptr;   // pinned
    size_t h_partials_cap = 0;
    hipEvent_t ev[10] = {};       // stage boundaries; ev[8] = accumulation done (stream A -> stream B dependency); ev[9] = accumulation starts
    hipStream_t stream_b = nullptr;   // B: this slot's bucket reduction + partials to host
    hipStream_t stream_c = nullptr;   // C: reduction of the second window group when a lone call is split in two
    hipEvent_t ev_c = nullptr;        // second group's reduction done
    std::vector<uint32_t> part_off, part_cnt;   // per owned window: first partial, number of partials
    bool pending = false;
    bool collecting = false;      // a thread is waiting for / combining this call
    bool sync_owned = false;      // submitted by a synchronous entry point: its own thread will collect it
    bool trivial = false;         // n == 0 / empty bit range / no owned window: the result is infinity, no GPU work queued
    uint64_t ticket = 0;
    Geom g{};
    uint32_t wgs_per_window = 0, nparts = 0, NB = 0;
    bool raw_out = false;
    int (*finish)(msm_amd_ctx*, Slot*, uint8_t*) = nullptr;
    DevBuf* bufs[21] = {&d_xbeta, &d_dig, &d_tmp, &d_M, &d_Mscan, &d_tailctl, &d_work, &d_biglist, &d_segout, &d_bases, &d_scalars,
                        &d_affine, &d_count, &d_offset, &d_cursor, &d_order, &d_bins, &d_tiles, &d_sorted, &d_buckets, &d_partials};
};

struct msm_amd_ctx {
    int device = 0;
    // Threads: submitting (GPU work is queued on shared streams) and collecting (host combine, shared timing fields) are
    // serialised by `mu`; WAITING for the GPU happens outside it, so calls issued from different threads -- e.g. the libuv
    // workers behind two unresolved JS promises -- overlap on the GPU like explicitly pipelined ones.  A synchronous call that
    // finds every slot taken sleeps on `cv` until one is collected.
    std::mutex mu;
    std::condition_variable cv;
    hipStream_t stream = nullptr;     // A: bucket accumulation (the ALU-bound stage), in call order
    hipStream_t stream_s = nullptr;   // S: transfers + bucket sort of the NEXT call, under the current call's accumulation
    hipStream_t stream_s2 = nullptr;  // S2: the same for calls whose accumulation is small (few owned windows of a sharded call, small n)
    static constexpr uint32_t MAX_SLOTS = 8;
    Slot slots[MAX_SLOTS];
    uint32_t nslots = 2;              // calls that may be in flight ("slots" option, 1..8)
    uint32_t next_slot = 0;
    uint64_t next_ticket = 1;
    std::string err;
    uint32_t force_c = 0;
    uint32_t force_seg = 0; // MSM_AMD_SEG: bucket segment length (tests force splitting with a small value)
    uint32_t force_L = 0;   // MSM_AMD_REDUCE_L: buckets per lane in k_bucket_reduce (tuning)
    int no_split2 = 0;      // option "split2" = 0 switches the two-lanes-per-bucket accumulation of small launches off
    uint32_t split2_nb = 2u * 1024u * 64u;   // option "split2_nb": launches with fewer buckets than this use it (default: two waves per SIMD)
    uint32_t max_slice = 1u << 26;   // synchronous calls with more points are cut into slices of this many (option "slice")
    int glv_mode = 1;       // option "glv": 0 = never, 1 = where it pays (BLS12-381 G1, n <= 2^17), 2 = whenever it applies
    int no_split = 1;       // option "split" = 1 splits a lone call into two window groups (measured slower: off by default)
    int sort_atomic = 0;    // MSM_AMD_SORT=atomic selects the global-atomics scatter (A/B comparisons)
    float timings[8] = {};
    uint32_t geometry[6] = {};
};

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                     \
            return MSM_AMD_EHIP;                                                                               \
        }                                                                                                      \
    } while (0)

static int ensure(msm_amd_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes && b.p) return 0;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

template <class F>
static void host_store_jacobian_normalised(const XYZZ<F>& r, uint8_t* out) {
    Affine<F> a = XYZZ<F>::to_affine(r);
    F one = F::one(), zero = F::zero();
    if (r.is_inf()) {
        memcpy(out, &zero, sizeof(F));
        memcpy(out + sizeof(F), &one, sizeof(F));
        memcpy(out + 2 * sizeof(F), &zero, sizeof(F));
    } else {
        memcpy(out, &a.x, sizeof(F));
        memcpy(out + sizeof(F), &a.y, sizeof(F));
        memcpy(out + 2 * sizeof(F), &one, sizeof(F));
    }
}
// un-normalised Jacobian image of an XYZZ point: (X*ZZ*ZZZ^2.. ) -- choose Z = ZZZ*ZZ? we use the
// identity x = X/ZZ = (X*ZZ)/ZZ^2... simplest exact map: Z = ZZ * ZZZ, X' = X * ZZ * ZZZ^2, Y' = Y * ZZ^3 * ZZZ^2.
// (Z^2 = ZZ^2 ZZZ^2 => X'/Z^2 = X/ZZ ; Z^3 = ZZ^3 ZZZ^3 => Y'/Z^3 = Y/ZZZ.)
template <class F>
static void host_store_jacobian_raw(const XYZZ<F>& r, uint8_t* out) {
    F one = F::one(), zero = F::zero();
    if (r.is_inf()) {
        memcpy(out, &zero, sizeof(F));
        memcpy(out + sizeof(F), &one, sizeof(F));
        memcpy(out + 2 * sizeof(F), &zero, sizeof(F));
        return;
    }
    F zzz2 = F::sqr(r.zzz);
    F x = F::mul(F::mul(r.x, r.zz), zzz2);
    F zz3 = F::mul(F::sqr(r.zz), r.zz);
    F y = F::mul(F::mul(r.y, zz3), zzz2);
    F z = F::mul(r.zz, r.zzz);
    memcpy(out, &x, sizeof(F));
    memcpy(out + sizeof(F), &y, sizeof(F));
    memcpy(out + 2 * sizeof(F), &z, sizeof(F));
}
template <class F>
static XYZZ<F> host_load_jacobian(const uint8_t* in) {
    Jacobian<F> j;
    memcpy(&j, in, sizeof(j));
    return XYZZ<F>::from_jacobian(j);
}

static uint32_t pick_window_bits(uint32_t n, uint32_t nbits, uint32_t forced) {
    uint32_t c;
    if (forced) {
        c = forced;
    } else {
        // measured on MI355X (profiles/r01_sweep_window_bits.log): the bucket reduction costs ~1.1 ms whatever n is
        // (latency-bound), so 2^15 buckets per window win as soon as n reaches a few thousand; below that, 8 bits.
        c = n >= 4096 ? 16 : 8;
    }
    if (c > nbits) c = nbits;
    if (c < 1) c = 1;
    if (c > 16) c = 16;   // 16-bit digit codes, 15-bit bucket index
    return c;
}

struct RunArgs {
    const void* bases;
    const void* scalars;
    uint32_t scalar_size, n, flags;
    uint32_t bit_lo, nbits;   // scalar bit range
    uint32_t shard, nshards;
    bool raw_out;             // un-normalised Jacobian partial (multi-GPU shard)
};

template <class F>
static int finish_msm(msm_amd_ctx* ctx, Slot* sl, uint8_t* out);

template <class F>
static int submit_msm(msm_amd_ctx* ctx, Slot* sl, const RunArgs& a) {
    using X = XYZZ<F>;
    using FK = typename KernelField<F>::type;
    const size_t ASZ = sizeof(Affine<F>), JSZ = sizeof(Jacobian<F>), XSZ = sizeof(X);   // packed images
    const size_t BSZ = bucket_bytes<FK>();                                               // one bucket in HBM
    sl->trivial = false;
    sl->raw_out = a.raw_out;
    sl->finish = &finish_msm<F>;
    if (a.n == 0 || a.nbits == 0) {
        sl->trivial = true;
        return 0;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // everything up to the sorted index lists runs on stream S -- or S2 when the accumulation will be small: such calls run their
    // accumulation and reduction on their slot's own stream, several of them in flight, and the ~ten short sort kernels of the next
    // call must not queue behind one of those in a shared hardware queue (8-way shard of n = 2^20: 0.77 -> 0.56 ms per call with
    // both measures; calls with a full-size accumulation were 3-8 % slower on S2, hence two streams)
    bool small_est;
    {
        const uint32_t c0 = pick_window_bits(a.n, a.nbits, ctx->force_c), w0 = a.nbits / c0 + 1;
        small_est = (uint64_t)((w0 + a.nshards - 1) / a.nshards) << (c0 - 1) < 2u * 1024u * 64u;
    }
    if (small_est && !ctx->stream_s2) {
        // created on first use: the mere existence of a stream of another priority re-partitions the hardware queues and cost calls
        // with a full-size accumulation ~5 % (measured); of another priority because only such streams never share a queue with S/A/B
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream_s2, hipStreamNonBlocking, 1));
    }
    hipStream_t st = small_est ? ctx->stream_s2 : ctx->stream_s;
    const bool dev_scalars = (a.flags & MSM_AMD_DEVICE_PTRS) != 0;
    const bool dev_bases = dev_scalars || (a.flags & MSM_AMD_DEVICE_BASES) != 0;
    const bool jac = (a.flags & MSM_AMD_BASES_JACOBIAN) != 0;
    const size_t base_bytes = (size_t)a.n * (jac ? JSZ : ASZ);
    const size_t scalar_bytes = (size_t)a.n * a.scalar_size;

    const uint8_t* d_bases;
    const uint8_t* d_scalars;
    if (dev_bases) {
        if (((uintptr_t)a.bases & 15) != 0) {
            ctx->err = "device bases pointer must be 16-byte aligned";
            return MSM_AMD_EINVAL;
        }
        d_bases = (const uint8_t*)a.bases;
    } else {
        if (int rc = ensure(ctx, sl->d_bases, base_bytes)) return rc;
        HIPCHK(ctx, hipMemcpyAsync(sl->d_bases.p, a.bases, base_bytes, hipMemcpyHostToDevice, st));
        d_bases = (const uint8_t*)sl->d_bases.p;
    }
    if (dev_scalars) {
        d_scalars = (const uint8_t*)a.scalars;
    } else {
        if (int rc = ensure(ctx, sl->d_scalars, scalar_bytes + 8)) return rc;
        HIPCHK(ctx, hipMemcpyAsync(sl->d_scalars.p, a.scalars, scalar_bytes, hipMemcpyHostToDevice, st));
        d_scalars = (const uint8_t*)sl->d_scalars.p;
    }

    Geom g;
    g.n = a.n;
    g.scalar_size = a.scalar_size;
    g.bit_lo = a.bit_lo;
    g.nbits = a.nbits;
    g.c = pick_window_bits(a.n, a.nbits, ctx->force_c);
    g.W = a.nbits / g.c + 1;
    g.half = 1u << (g.c - 1);
    g.shard = a.shard;
    g.nshards = a.nshards;
    // GLV path: 2n virtual items with 127-bit signed halves, 8 windows of 16 bits (k_digits_glv).  Same number of bucket
    // additions, half the buckets.  It pays where the bucket reduction dominates (n <= 2^17: 8-15 % per call); from 2^18 up
    // the accumulation dominates and 8 x 2^15 buckets give only 4096 accumulation waves, which quantise badly over
    // 1024 SIMDs x 3 waves (measured 3-15 % slower), so larger calls keep the 16-window schedule unless option "glv" = 2
    // forces the path (tests, experiments).
    bool use_glv = false;
    if constexpr (GlvField<F>::value)
        use_glv = ctx->glv_mode != 0 && !ctx->sort_atomic && g.c == 16 && a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256 && a.n >= 1024 &&
                  (a.n <= (1u << 17) || ctx->glv_mode == 2) && a.n <= (1u << 26) && (((uintptr_t)d_scalars & 15) == 0);
    const uint32_t nv = use_glv ? 2 * a.n : a.n;   // items the sort and the accumulation see
    if (use_glv) {
        g.n = nv;
        g.nbits = 128;
        g.W = 8;
    }
    g.Wl = 0;
    for (uint32_t w = 0; w < g.W; ++w)
        if (w % g.nshards == g.shard) ++g.Wl;
    ctx->geometry[0] = g.c;
    ctx->geometry[1] = g.W;
    ctx->geometry[2] = g.half;
    ctx->geometry[3] = g.Wl;
    ctx->geometry[4] = ctx->geometry[5] = 0;
    sl->g = g;
    if (g.Wl == 0) {
        sl->trivial = true;
        return 0;
    }
    const uint32_t NB = g.Wl * g.half;
    const bool fast = (a.scalar_size % 4 == 0) && (((uintptr_t)d_scalars & 3) == 0);

    HIPCHK(ctx, hipEventRecord(sl->ev[0], st));
    // ---- Jacobian bases -> affine
    if (jac) {
        if (int rc = ensure(ctx, sl->d_affine, (size_t)a.n * ASZ)) return rc;
        uint32_t lanes = (a.n + 7) / 8;
        k_jacobian_to_affine<F><<<(lanes + 255) / 256, 256, 0, st>>>(d_bases, a.n, (uint8_t*)sl->d_affine.p);
        d_bases = (const uint8_t*)sl->d_affine.p;
    }
    const uint8_t* d_xbeta = nullptr;
    if constexpr (GlvField<F>::value) {
        if (use_glv) {
            if (int rc = ensure(ctx, sl->d_xbeta, (size_t)a.n * sizeof(F))) return rc;
            k_glv_xbeta<F><<<(a.n + 255) / 256, 256, 0, st>>>(d_bases, a.n, (uint8_t*)sl->d_xbeta.p);
            d_xbeta = (const uint8_t*)sl->d_xbeta.p;
        }
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[1], st));

    const uint32_t ntiles = (NB + SCAN_TILE - 1) / SCAN_TILE;
    if (int rc = ensure(ctx, sl->d_count, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_offset, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_cursor, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_order, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_bins, ORDER_KEYS * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_tiles, (size_t)ntiles * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_sorted, (size_t)g.Wl * nv * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_buckets, (size_t)NB * BSZ)) return rc;
    uint32_t* count = (uint32_t*)sl->d_count.p;
    uint32_t* offset = (uint32_t*)sl->d_offset.p;
    uint32_t* cursor = (uint32_t*)sl->d_cursor.p;
    uint32_t* order = (uint32_t*)sl->d_order.p;
    uint32_t* bins = (uint32_t*)sl->d_bins.p;
    uint32_t* tiles = (uint32_t*)sl->d_tiles.p;
    uint32_t* sorted = (uint32_t*)sl->d_sorted.p;

    const uint32_t nblk = (a.n + 255) / 256;
    const uint32_t oblk = std::max(1u, std::min(256u, NB / 8192u));
    // A lone call (nothing else in flight) is latency-bound at its end: the bucket reduction is ~40 dependent additions deep
    // on one wave per SIMD.  Optionally (option "split") its windows are handled as two groups, the reduction of the first
    // group running under the accumulation of the second.  Measured at n = 2^20 this LOSES (6.3 vs 4.8 ms): the 256-VGPR
    // reduction waves leave room for only one accumulation wave per SIMD, and two launches have two tails.  Kept for
    // experiments, off by default; overlapping across calls (slots) is what pays.
    bool others_pending = false;
    for (uint32_t k = 0; k < ctx->nslots; ++k) others_pending |= (&ctx->slots[k] != sl && ctx->slots[k].pending);
    const bool split_mode = !others_pending && !ctx->no_split && g.Wl >= 6 && NB >= 8u * 65536u;
    const uint32_t w_split = split_mode ? g.Wl - std::max(2u, (g.Wl * 3u) / 10u) : g.Wl;   // windows in the first group
    const uint32_t split = w_split * g.half;                                               // first bucket of the second group
    HIPCHK(ctx, hipMemsetAsync(bins, 0, ORDER_KEYS * 4, st));
    if (ctx->sort_atomic) {
        // ---- 1. digits + histogram (global atomics)
        HIPCHK(ctx, hipMemsetAsync(count, 0, (size_t)NB * 4, st));
        if (fast) k_histogram<true><<<nblk, 256, 0, st>>>(d_scalars, g, count);
        else k_histogram<false><<<nblk, 256, 0, st>>>(d_scalars, g, count);
        HIPCHK(ctx, hipEventRecord(sl->ev[2], st));
        // ---- 2. scan + order
        k_scan_tile_sums<<<ntiles, SCAN_BLOCK, 0, st>>>(count, NB, tiles);
        k_scan_tiles<<<1, SCAN_BLOCK, 0, st>>>(tiles, ntiles);
        k_scan_apply<<<ntiles, SCAN_BLOCK, 0, st>>>(count, NB, tiles, offset, cursor);
        k_order_hist<<<oblk, 1024, 0, st>>>(count, NB, split, bins);
        k_order_scan<<<1, 1024, 0, st>>>(bins);
        k_order_scatter<<<oblk, 1024, 0, st>>>(count, NB, split, bins, order);
        HIPCHK(ctx, hipEventRecord(sl->ev[3], st));
        // ---- 3. scatter
        if (fast) k_scatter<true><<<nblk, 256, 0, st>>>(d_scalars, g, cursor, sorted);
        else k_scatter<false><<<nblk, 256, 0, st>>>(d_scalars, g, cursor, sorted);
        HIPCHK(ctx, hipEventRecord(sl->ev[4], st));
    } else {
        // ---- 1. digit codes + coarse histogram
        const uint32_t fine_bits = std::min(8u, g.c - 1);
        const uint32_t cb = g.half >> fine_bits;                      // coarse bins per window (<= 128)
        const uint32_t stile = (nv + SORT_TILE - 1) / SORT_TILE;
        const size_t NM = (size_t)g.Wl * cb * stile;                   // count matrix entries (+1 for the total)
        const uint32_t mtiles = (uint32_t)((NM + 1 + SCAN_TILE - 1) / SCAN_TILE);
        if (int rc = ensure(ctx, sl->d_dig, (size_t)g.Wl * nv * 2)) return rc;
        if (int rc = ensure(ctx, sl->d_tmp, (size_t)g.Wl * nv * 8)) return rc;
        if (int rc = ensure(ctx, sl->d_M, (NM + 1) * 4)) return rc;
        if (int rc = ensure(ctx, sl->d_Mscan, (NM + 1) * 4 * 2)) return rc;
        if (int rc = ensure(ctx, sl->d_tiles, (size_t)std::max(mtiles, ntiles) * 4)) return rc;
        tiles = (uint32_t*)sl->d_tiles.p;
        uint16_t* dig = (uint16_t*)sl->d_dig.p;
        uint64_t* tmp = (uint64_t*)sl->d_tmp.p;
        uint32_t* M = (uint32_t*)sl->d_M.p;
        uint32_t* Mscan = (uint32_t*)sl->d_Mscan.p;
        uint32_t* Mdummy = Mscan + (NM + 1);
        const bool c16 = g.c == 16 && a.scalar_size == 32 && g.bit_lo == 0 && g.nbits == 256 && g.W == 17 && (((uintptr_t)d_scalars & 15) == 0);
        if (use_glv) k_digits_glv<<<nblk, 256, 0, st>>>(d_scalars, a.n, g, dig);
        else if (c16) k_digits_c16<<<nblk, 256, 0, st>>>(d_scalars, g, dig);
        else if (fast) k_digits<true><<<nblk, 256, 0, st>>>(d_scalars, g, dig);
        else k_digits<false><<<nblk, 256, 0, st>>>(d_scalars, g, dig);
        HIPCHK(ctx, hipMemsetAsync(M + NM, 0, 4, st));
        k_coarse_hist<<<dim3(stile, g.Wl), SORT_BLOCK, 0, st>>>(dig, nv, fine_bits, cb, stile, M);
        HIPCHK(ctx, hipEventRecord(sl->ev[2], st));
        // ---- 2. scan of the count matrix
        k_scan_tile_sums<<<mtiles, SCAN_BLOCK, 0, st>>>(M, (uint32_t)(NM + 1), tiles);
        k_scan_tiles<<<1, SCAN_BLOCK, 0, st>>>(tiles, mtiles);
        k_scan_apply<<<mtiles, SCAN_BLOCK, 0, st>>>(M, (uint32_t)(NM + 1), tiles, Mscan, Mdummy);
        HIPCHK(ctx, hipEventRecord(sl->ev[3], st));
        // ---- 3. coarse scatter, fine sort (emits count/offset), ordering by population
        k_coarse_scatter<<<dim3(stile, g.Wl), SORT_BLOCK, 0, st>>>(dig, nv, fine_bits, cb, stile, Mscan, tmp);
        k_fine_sort<<<g.Wl * cb, SORT_BLOCK, 0, st>>>(tmp, Mscan, stile, g.Wl * cb, fine_bits, cb, g.half, sorted, count, offset);
        k_order_hist<<<oblk, 1024, 0, st>>>(count, NB, split, bins);
        k_order_scan<<<1, 1024, 0, st>>>(bins);
        k_order_scatter<<<oblk, 1024, 0, st>>>(count, NB, split, bins, order);
        HIPCHK(ctx, hipEventRecord(sl->ev[4], st));
    }
    // ---- 4. accumulate (stream A, after this call's sort and after the previous call's accumulation).
    // A SMALL accumulation (fewer than two waves per SIMD: few owned windows of a sharded call, small n) cannot fill the
    // chip by itself, so it goes to the slot's own stream instead and may overlap the accumulations of the other calls in flight.
    const bool small_acc = NB < 2u * 1024u * 64u && !split_mode;
    st = small_acc ? sl->stream_b : ctx->stream;   // (the slot's B stream: its reduction follows on the same stream anyway)
    HIPCHK(ctx, hipStreamWaitEvent(st, sl->ev[4], 0));
    HIPCHK(ctx, hipEventRecord(sl->ev[9], st));
    // segment length: mean bucket population + 8 standard deviations of a Poisson bucket, at least 32: random inputs
    // practically never exceed it (a handful of segments queued for the tail kernels costs a serial chain of `seg` additions
    // on one lane, ~0.6 ms at seg = 32, which a 2x-the-mean rule paid at mean = 16).  Longer buckets -- adversarial scalars, or
    // the few buckets of a narrow top window when the scalar width is not a multiple of c -- are cut into segments, which
    // bounds the longest dependent chain of additions.
    const uint32_t mean = (nv + g.half - 1) / g.half;
    uint32_t seg = ctx->force_seg ? ctx->force_seg : std::max(32u, mean + 8u * (uint32_t)std::ceil(std::sqrt((double)mean)) + 8u);
    const size_t max_segs = (size_t)g.Wl * nv / seg + 1;
    if (int rc = ensure(ctx, sl->d_tailctl, sizeof(uint32_t) * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_work, max_segs * 8)) return rc;
    if (int rc = ensure(ctx, sl->d_biglist, max_segs * 16)) return rc;
    if (int rc = ensure(ctx, sl->d_segout, max_segs * BSZ)) return rc;
    HIPCHK(ctx, hipMemsetAsync(sl->d_tailctl.p, 0, 16, st));
    uint32_t tail_block = 256;
    while ((size_t)tail_block * BSZ > 64 * 1024 && tail_block > 64) tail_block /= 2;
    uint8_t* buckets = (uint8_t*)sl->d_buckets.p;
    // G1 has two builds of the accumulation kernel (bucket_kernels.hpp): the chained one when the launch puts at least two waves
    // on every SIMD, the two-chain one otherwise
    auto accumulate = [&](const uint32_t* ord, uint32_t nbk) {
        if constexpr (KernelField<F>::chained_variant) {
            // few buckets with enough entries each (the windows of one rank of a sharded call): two lanes per bucket
            if (!ctx->no_split2 && nbk < ctx->split2_nb && mean >= 16) {
                launch_accumulate_split2<FK>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
                return;
            }
            if ((uint64_t)nbk >= 2u * 1024u * 64u) {
                launch_accumulate<FK, 1>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
                return;
            }
        }
        launch_accumulate<FK, 0>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
    };
    accumulate(order, split);
    launch_accumulate_tail<FK>(st, d_bases, d_xbeta, sorted, offset, count, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p,
                               (uint8_t*)sl->d_segout.p, tail_block);
    HIPCHK(ctx, hipEventRecord(sl->ev[8], st));   // first group accumulated
    if (split < NB) {
        HIPCHK(ctx, hipMemsetAsync(sl->d_tailctl.p, 0, 16, st));
        accumulate(order + split, NB - split);
        launch_accumulate_tail<FK>(st, d_bases, d_xbeta, sorted, offset, count, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p,
                                   (uint8_t*)sl->d_segout.p, tail_block);
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[5], st));
    // ---- 5. reduce (streams B/C: overlap the next call's, or the second group's, accumulation on stream A)
    hipStream_t sb = sl->stream_b;
    // buckets per lane: up to 8 (16 where a point takes two threads, so that the waves still fit one per SIMD) for a lone call (its reduction is on the critical path: short dependent chains, one wave on
    // every SIMD); 32 when another call is already in flight: the reduction then hides behind that call's accumulation, and
    // what counts is its SIMD time and its register footprint -- fewer, longer lanes issue fewer instructions in total
    // (the double-and-add by the lane's first index is paid once per lane) and leave 3/4 of the SIMDs to the
    // accumulation waves alone (n = 2^20, two calls in flight: 3.43 ms per call with L = 8, 3.09 ms with L = 32,
    // 3.21 ms with L = 64 where the reduction outlasts the accumulation it hides behind)
    struct Plan {
        uint32_t L, lanes, rblock, wgs;
    };
    auto plan = [&](uint32_t nbuckets) {
        Plan p;
        p.L = others_pending ? std::min(32u, std::max(1u, (uint32_t)(((uint64_t)g.Wl * g.half) / 16384u)))
                             : std::min(8u * lanes_per_point<FK>(), std::max(1u, (nbuckets * lanes_per_point<FK>() + 65535u) / 65536u));
        // a power of two (the nearer one): a lane multiplies its run by its first index t*L with double-and-add, and the zero low
        // bits of that index cost no additions (sharded calls: L = 10 -> 8 took 1.25 -> 1.17 ms per call at 4 ranks, 18 -> 16 1.62 -> 1.56 at 2)
        {
            uint32_t f = 1;
            while (f * 2 <= p.L) f *= 2;
            p.L = (2 * p.L >= 3 * f) ? 2 * f : f;
        }
        if (ctx->force_L) p.L = ctx->force_L;
        p.lanes = (g.half + p.L - 1) / p.L;
        p.rblock = std::min(256u, std::max(64u, p.lanes));
        while ((size_t)p.rblock * BSZ > 64 * 1024 && p.rblock > 64) p.rblock /= 2;   // LDS budget (64 KB default limit)
        p.wgs = (p.lanes + p.rblock - 1) / p.rblock;
        return p;
    };
    const Plan p0 = plan(split), p1 = plan(NB - split);
    const uint32_t nparts0 = w_split * p0.wgs, nparts1 = (g.Wl - w_split) * p1.wgs, nparts = nparts0 + nparts1;
    sl->part_off.assign(g.Wl, 0);
    sl->part_cnt.assign(g.Wl, 0);
    for (uint32_t wl = 0; wl < g.Wl; ++wl) {
        sl->part_off[wl] = wl < w_split ? wl * p0.wgs : nparts0 + (wl - w_split) * p1.wgs;
        sl->part_cnt[wl] = wl < w_split ? p0.wgs : p1.wgs;
    }
    if (int rc = ensure(ctx, sl->d_partials, (size_t)nparts * XSZ)) return rc;
    if (sl->h_partials_cap < (size_t)nparts * XSZ + 16) {
        if (sl->h_partials) HIPCHK(ctx, hipHostFree(sl->h_partials));
        sl->h_partials = nullptr;
        sl->h_partials_cap = 0;
        HIPCHK(ctx, hipHostMalloc(&sl->h_partials, (size_t)nparts * XSZ * 2 + 64, hipHostMallocDefault));
        sl->h_partials_cap = (size_t)nparts * XSZ * 2 + 64;
    }
    uint8_t* partials = (uint8_t*)sl->d_partials.p;
    HIPCHK(ctx, hipStreamWaitEvent(sb, sl->ev[8], 0));
    launch_bucket_reduce<FK>(sb, nparts0, p0.rblock, buckets, g.half, p0.L, p0.lanes, p0.wgs, partials, offset, count, 0, NB);
    if (nparts1) {
        hipStream_t sc = sl->stream_c;
        HIPCHK(ctx, hipStreamWaitEvent(sc, sl->ev[5], 0));
        launch_bucket_reduce<FK>(sc, nparts1, p1.rblock, buckets + (size_t)split * BSZ, g.half, p1.L, p1.lanes, p1.wgs, partials + (size_t)nparts0 * XSZ, offset,
                                 count, split, NB);
        HIPCHK(ctx, hipEventRecord(sl->ev_c, sc));
        HIPCHK(ctx, hipStreamWaitEvent(sb, sl->ev_c, 0));
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[6], sb));
    HIPCHK(ctx, hipMemcpyAsync(sl->h_partials, sl->d_partials.p, (size_t)nparts * XSZ, hipMemcpyDeviceToHost, sb));
    // total number of scattered (pair, window) entries = last offset + last count (for the roofline accounting)
    uint32_t* h_tail = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(sl->h_partials) + (size_t)nparts * XSZ);
    HIPCHK(ctx, hipMemcpyAsync(h_tail, offset + (NB - 1), 4, hipMemcpyDeviceToHost, sb));
    HIPCHK(ctx, hipMemcpyAsync(h_tail + 1, count + (NB - 1), 4, hipMemcpyDeviceToHost, sb));
    HIPCHK(ctx, hipEventRecord(sl->ev[7], sb));
    const uint32_t wgs_per_window = p0.wgs;
    sl->wgs_per_window = wgs_per_window;
    sl->nparts = nparts;
    sl->NB = NB;
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

template <class F>
static int finish_msm(msm_amd_ctx* ctx, Slot* sl, uint8_t* out) {
    using X = XYZZ<F>;
    const size_t XSZ = sizeof(X);
    for (float& t : ctx->timings) t = 0.f;
    const Geom& g = sl->g;
    ctx->geometry[0] = g.c;
    ctx->geometry[1] = g.W;
    ctx->geometry[2] = g.half;
    ctx->geometry[3] = g.Wl;
    ctx->geometry[4] = ctx->geometry[5] = 0;
    if (sl->trivial) {
        host_store_jacobian_normalised(XYZZ<typename HostOf<F>::type>::inf(), out);
        return 0;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(sl->ev[7]));
    const uint32_t nparts = sl->nparts;
    uint32_t* h_tail = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(sl->h_partials) + (size_t)nparts * XSZ);
    {
        uint64_t entries = (uint64_t)h_tail[0] + h_tail[1];
        ctx->geometry[4] = (uint32_t)entries;
        ctx->geometry[5] = (uint32_t)(entries >> 32);
    }

    // ---- 6. host combine: window sums, Horner over owned windows (top first), each weighted 2^(c*w)
    auto t0 = std::chrono::steady_clock::now();
    using HF = typename HostOf<F>::type;
    using HX = XYZZ<HF>;
    static_assert(sizeof(HX) == sizeof(X), "host twin has the same byte image");
    const HX* parts = reinterpret_cast<const HX*>(sl->h_partials);
    HX total = HX::inf();
    int prev_w = -1;   // global index of the window most recently folded in
    for (int wl = (int)g.Wl - 1; wl >= 0; --wl) {
        int wglob = wl * (int)g.nshards + (int)g.shard;
        if (prev_w >= 0)
            for (int k = 0; k < (prev_w - wglob) * (int)g.c; ++k) total = HX::dbl(total);
        HX s = HX::inf();
        for (uint32_t k = 0; k < sl->part_cnt[wl]; ++k) s = HX::add(s, parts[(size_t)sl->part_off[wl] + k]);
        total = HX::add(total, s);
        prev_w = wglob;
    }
    for (int k = 0; k < prev_w * (int)g.c; ++k) total = HX::dbl(total);
    if (sl->raw_out) host_store_jacobian_raw(total, out);
    else host_store_jacobian_normalised(total, out);
    auto t1 = std::chrono::steady_clock::now();

    float ms;
    (void)hipEventElapsedTime(&ms, sl->ev[0], sl->ev[7]); ctx->timings[0] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[1], sl->ev[2]); ctx->timings[1] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[2], sl->ev[3]); ctx->timings[2] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[3], sl->ev[4]); ctx->timings[3] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[9], sl->ev[5]); ctx->timings[4] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[5], sl->ev[6]); ctx->timings[5] = ms;
    ctx->timings[6] = std::chrono::duration<float, std::milli>(t1 - t0).count();
    (void)hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]); ctx->timings[7] = ms;
    return 0;
}

static int check_args(msm_amd_ctx* ctx, const RunArgs& a) {
    if (a.n && (!a.bases || !a.scalars)) {
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    if (a.scalar_size < 1 || a.scalar_size > 64) {
        ctx->err = "scalar_size must be in [1, 64]";
        return MSM_AMD_EINVAL;
    }
    if (a.n > (1u << 27)) {   // (pair, window) positions are 32-bit: 17 windows x n must stay below 2^32
        ctx->err = "n must not exceed 2^27 per asynchronous call (the synchronous entry points slice larger inputs themselves)";
        return MSM_AMD_EINVAL;
    }
    if (a.nshards == 0 || a.shard >= a.nshards) {
        ctx->err = "bad shard";
        return MSM_AMD_EINVAL;
    }
    return 0;
}

// Queue one MSM on the next slot; the caller collects it with finish_ticket().
static int submit(msm_amd_ctx* ctx, int curve, int group, const RunArgs& a, uint64_t* ticket, bool wait_for_slot = false) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (int rc = check_args(ctx, a)) return rc;
    Slot* sl = nullptr;
    for (;;) {
        for (uint32_t k = 0; k < ctx->nslots; ++k) {   // tickets may be collected in any order: take any free slot
            Slot* cand = &ctx->slots[(ctx->next_slot + k) % ctx->nslots];
            if (!cand->pending) {
                sl = cand;
                break;
            }
        }
        if (sl || !wait_for_slot) break;
        // every slot is taken.  Slots held by synchronous calls of other threads will be collected by those threads: wait for one.
        // If all of them hold asynchronous tickets, only the caller can free one -- waiting would never end.
        bool some_sync = false;
        for (uint32_t k = 0; k < ctx->nslots; ++k) some_sync |= ctx->slots[k].sync_owned;
        if (!some_sync) break;
        ctx->cv.wait(lk);
    }
    if (!sl) {
        ctx->err = "every slot of this context holds an uncollected call: wait for an older ticket first";
        return MSM_AMD_EINVAL;
    }
    int rc;
    if (curve == MSM_AMD_BLS12381 && group == 1) rc = submit_msm<Fp<Bls12381Fp>>(ctx, sl, a);
    else if (curve == MSM_AMD_BLS12381 && group == 2) rc = submit_msm<Fp2<Bls12381Fp>>(ctx, sl, a);
    else if (curve == MSM_AMD_BN128 && group == 1) rc = submit_msm<Fp<Bn254Fp>>(ctx, sl, a);
    else if (curve == MSM_AMD_BN128 && group == 2) rc = submit_msm<Fp2<Bn254Fp>>(ctx, sl, a);
    else {
        ctx->err = "unknown curve/group";
        return MSM_AMD_EINVAL;
    }
    if (rc) {
        // leave the streams clean so the context stays usable
        (void)hipStreamSynchronize(ctx->stream_s);
        if (ctx->stream_s2) (void)hipStreamSynchronize(ctx->stream_s2);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(sl->stream_c);
        (void)hipStreamSynchronize(sl->stream_b);
        return rc;
    }
    sl->pending = true;
    sl->sync_owned = wait_for_slot;
    sl->ticket = ctx->next_ticket++;
    *ticket = sl->ticket;
    ctx->next_slot = ((uint32_t)(sl - ctx->slots) + 1u) % ctx->nslots;
    return 0;
}

static int finish_ticket(msm_amd_ctx* ctx, uint64_t ticket, uint8_t* out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    Slot* found = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (Slot& sl : ctx->slots)
            if (sl.pending && !sl.collecting && sl.ticket == ticket) {
                found = &sl;
                sl.collecting = true;
                break;
            }
        if (!found) {
            ctx->err = "unknown or already collected ticket";
            return MSM_AMD_EINVAL;
        }
    }
    // wait for the GPU outside the lock (other threads may submit and collect meanwhile)
    if (!found->trivial) {
        (void)hipSetDevice(ctx->device);
        (void)hipEventSynchronize(found->ev[7]);
    }
    int rc;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        rc = found->finish(ctx, found, out);
        found->pending = false;
        found->collecting = false;
        found->sync_owned = false;
    }
    ctx->cv.notify_all();
    return rc;
}

static size_t elem_bytes(int curve, int group) { return (size_t)(curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group; }
static int dispatch(msm_amd_ctx* ctx, int curve, int group, const RunArgs& a, uint8_t* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (!out) {
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    if (a.n <= ctx->max_slice) {
        uint64_t t = 0;
        if (int rc = submit(ctx, curve, group, a, &t, true)) return rc;
        return finish_ticket(ctx, t, out);
    }
    // More points than one pass takes (the reference's external caller slices at 2^22 for its own reasons): the MSM is linear in the
    // points, so run slice after slice -- two in flight -- and add the partial results on the host.  Works for host and for device
    // pointers alike (plain address arithmetic); a shard call stays a shard call per slice.
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || !a.bases || !a.scalars) {
        ctx->err = "bad argument";
        return MSM_AMD_EINVAL;
    }
    const size_t bstride = ((a.flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * elem_bytes(curve, group), jsz = 3 * elem_bytes(curve, group);
    std::vector<uint8_t> parts;
    uint64_t pending_ticket = 0;
    bool have_pending = false;
    int rc = 0;
    for (uint64_t off = 0; off < a.n && rc == 0; off += ctx->max_slice) {
        RunArgs s = a;
        s.n = (uint32_t)std::min<uint64_t>(ctx->max_slice, a.n - off);
        s.bases = (const uint8_t*)a.bases + off * bstride;
        s.scalars = (const uint8_t*)a.scalars + off * a.scalar_size;
        s.raw_out = true;   // un-normalised partial: one inversion at the very end
        bool free_slot = false;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            for (uint32_t k = 0; k < ctx->nslots; ++k) free_slot |= !ctx->slots[k].pending;
        }
        if (have_pending && !free_slot) {   // nothing free (a single slot, or the others are taken): collect our own slice first
            parts.resize(parts.size() + jsz);
            rc = finish_ticket(ctx, pending_ticket, parts.data() + parts.size() - jsz);
            have_pending = false;
            if (rc) break;
        }
        uint64_t t = 0;
        rc = submit(ctx, curve, group, s, &t, true);
        if (rc) break;
        if (have_pending) {
            parts.resize(parts.size() + jsz);
            rc = finish_ticket(ctx, pending_ticket, parts.data() + parts.size() - jsz);
        }
        pending_ticket = t;
        have_pending = true;
    }
    if (have_pending) {
        parts.resize(parts.size() + jsz);
        int rc2 = finish_ticket(ctx, pending_ticket, parts.data() + parts.size() - jsz);
        if (!rc) rc = rc2;
    }
    if (rc) return rc;
    return msm_amd_combine(curve, group, parts.data(), (uint32_t)(parts.size() / jsz), out);   // normalised (for a shard too: any representative will do)
}

template <class F>
static int gen_points_t(msm_amd_ctx* ctx, const Affine<F>& gen, uint64_t seed, uint32_t n, void* dev_out) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    k_gen_points<F><<<(n + 255) / 256, 256, 0, ctx->stream>>>(gen, seed, n, (uint8_t*)dev_out);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

template <class F>
static int combine_t(const uint8_t* partials, uint32_t count, uint8_t* out) {
    XYZZ<F> total = XYZZ<F>::inf();
    for (uint32_t i = 0; i < count; ++i) total = XYZZ<F>::add(total, host_load_jacobian<F>(partials + (size_t)i * sizeof(Jacobian<F>)));
    host_store_jacobian_normalised(total, out);
    return 0;
}
template <class F, bool TO>
static int f_conv(const uint8_t* in, uint32_t count, uint8_t* out) {
    for (uint32_t i = 0; i < count; ++i) {
        F x;
        memcpy(&x, in + (size_t)i * sizeof(F), sizeof(F));
        F r = TO ? F::to_mont(x) : F::from_mont(x);
        memcpy(out + (size_t)i * sizeof(F), &r, sizeof(F));
    }
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
            ctx->err = "device pointers must be 16-byte aligned";
            return MSM_AMD_EINVAL;
        }
    } else {
        HIPCHK(ctx, hipMalloc(&di, in_b));
        if (hipMalloc(&dout, out_b) != hipSuccess) {
            (void)hipFree(di);
            ctx->err = "hipMalloc failed";
            return MSM_AMD_EHIP;
        }
        (void)hipMemcpyAsync(di, in, in_b, hipMemcpyHostToDevice, st);
    }
    const uint32_t blocks = (n + 255) / 256;
    if (which == 0) k_jacobian_to_affine<F><<<((n + 7) / 8 + 255) / 256, 256, 0, st>>>((const uint8_t*)di, n, (uint8_t*)dout);
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
        ctx->err = std::string("batch conversion: ") + hipGetErrorString(e);
        return MSM_AMD_EHIP;
    }
    if (hbad) {
        ctx->err = "CtoLEM: x^3 + b is not a square for " + std::to_string(hbad) + " element(s) (written as infinity; the reference does not terminate on such input)";
        return MSM_AMD_EINVAL;
    }
    return 0;
}
template <class P>
static Fp<P> fp_from(const uint32_t* limbs) {
    Fp<P> r;
    for (int i = 0; i < P::N; ++i) r.v[i] = limbs[i];
    return r;
}
static int conv(msm_amd_ctx* ctx, int which, int curve, int group, uint32_t flags, const void* in, uint32_t n, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out) {
        ctx->err = "null pointer";
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
    ctx->err = "unknown curve/group";
    return MSM_AMD_EINVAL;
}


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
        ctx->err = "device pointers must be 16-byte aligned";
        return MSM_AMD_EINVAL;
    }
    HIPCHK(ctx, hipMalloc(&dkey, 64));
    hipError_t e = hipSuccess;
    if (!dev) {
        if (hipMalloc(&di, in_b) != hipSuccess || hipMalloc(&dout, out_b) != hipSuccess) {
            ctx->err = "hipMalloc failed";
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
        ctx->err = std::string("batch_apply_key: ") + hipGetErrorString(e);
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
        ctx->err = std::string("fft: ") + hipGetErrorString(e);
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
        ctx->err = "device pointers must be 16-byte aligned";
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
        ctx->err = std::string("fft: ") + hipGetErrorString(e);
        return MSM_AMD_EHIP;
    }
    return 0;
}
template <class F>
static int in_curve_t(msm_amd_ctx* ctx, const F& b, uint32_t flags, const void* points, uint32_t n, uint8_t* out, bool group_check = false) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const bool dev = (flags & MSM_AMD_DEVICE_PTRS) != 0;
    const size_t in_b = (size_t)n * sizeof(Affine<F>);
    void *di = const_cast<void*>(points), *dout = out;
    if (dev) {
        if ((uintptr_t)points & 15) {
            ctx->err = "device points pointer must be 16-byte aligned";
            return MSM_AMD_EINVAL;
        }
    } else {
        HIPCHK(ctx, hipMalloc(&di, in_b));
        if (hipMalloc(&dout, n) != hipSuccess) {
            (void)hipFree(di);
            ctx->err = "hipMalloc failed";
            return MSM_AMD_EHIP;
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
        ctx->err = std::string("batch_in_curve_affine: ") + hipGetErrorString(e);
        return MSM_AMD_EHIP;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

// One wave that spins for `ticks` of the 100-MHz wall clock: the probe of the stream calibration below and of
// msm_amd_debug_stream_sharing.
__global__ void k_spin(uint64_t ticks) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// true when kernels on a and b serialise, i.e. the two streams were mapped onto the same hardware queue
static bool streams_serialise(hipStream_t a, hipStream_t b) {
    const uint64_t ticks = 15000;   // 0.15 ms each: ~0.17 ms together on different queues, ~0.32 ms on a shared one
    for (int attempt = 0; attempt < 2; ++attempt) {
        auto t0 = std::chrono::steady_clock::now();
        k_spin<<<1, 64, 0, a>>>(ticks);
        k_spin<<<1, 64, 0, b>>>(ticks);
        if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return false;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < 245.0) return false;   // concurrent (a slow verdict is taken twice before it is believed)
    }
    return true;
}

// HIP maps the streams of a process onto a few hardware queues (four by default), by creation order and by what the process created
// before -- and streams on one queue serialise.  Which of the engine's streams share a queue decides its throughput (n = 2^20, two
// calls in flight: 2.7-2.9 ms per call when A, S, B0 and B1 sit on four queues; 3.1-3.3 when A shares with S; 3.4-3.5 when A shares
// with B1; 3.6-3.75 when S shares with a B: profiles/r01_stream_queue_layouts.log).  So the context does not trust the creation
// order: it finds the queue classes of the streams it created (probe above, ~10 ms) and deals them out so that the streams that work
// side by side sit on different queues: A, S, B0, B1 on four classes; B2 with A and B3 with S (when four sharded calls are in flight
// A and S idle: their accumulations run on the B streams, their sorts on S2).  MSM_AMD_NO_CALIBRATE=1 keeps the creation order.
static void calibrate_streams(msm_amd_ctx* ctx) {
    if (getenv("MSM_AMD_NO_CALIBRATE")) return;
    std::vector<hipStream_t> all = {ctx->stream, ctx->stream_s};
    for (Slot& sl : ctx->slots) {
        all.push_back(sl.stream_b);
        all.push_back(sl.stream_c);
    }
    k_spin<<<1, 64, 0, all[0]>>>(100);   // loads the code object before anything is timed
    if (hipDeviceSynchronize() != hipSuccess) return;
    std::vector<std::vector<hipStream_t>> cls;
    for (hipStream_t st : all) {
        bool placed = false;
        for (auto& c : cls)
            if (streams_serialise(c[0], st)) {
                c.push_back(st);
                placed = true;
                break;
            }
        if (!placed) cls.push_back({st});
    }
    (void)hipGetLastError();
    if (cls.size() < 2) return;
    const size_t nc = cls.size();
    std::vector<hipStream_t> rest;
    auto take = [&](size_t c) -> hipStream_t {
        for (size_t k = 0; k < nc; ++k) {   // the wanted class, else the next one that still has a stream
            auto& v = cls[(c + k) % nc];
            if (!v.empty()) {
                hipStream_t st = v.back();
                v.pop_back();
                return st;
            }
        }
        return nullptr;
    };
    ctx->stream = take(0);
    ctx->stream_s = take(1);
    const size_t nslots = sizeof(ctx->slots) / sizeof(ctx->slots[0]);
    for (size_t k = 0; k < nslots; ++k) ctx->slots[k].stream_b = take((2 + k) % nc);
    for (size_t k = 0; k < nslots; ++k) ctx->slots[k].stream_c = take((2 + k) % nc);
}

int msm_amd_create(int device_id, msm_amd_ctx** out_ctx) {
    if (!out_ctx) return MSM_AMD_EINVAL;
    *out_ctx = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device_id < 0 || device_id >= cnt) return MSM_AMD_EHIP;
    msm_amd_ctx* ctx = new (std::nothrow) msm_amd_ctx();
    if (!ctx) return MSM_AMD_ENOMEM;
    ctx->device = device_id;
    if (const char* e = getenv("MSM_AMD_REDUCE_L")) ctx->force_L = (uint32_t)atoi(e);
    if (const char* e = getenv("MSM_AMD_SEG")) ctx->force_seg = (uint32_t)atoi(e);
    if (const char* e = getenv("MSM_AMD_C")) ctx->force_c = (uint32_t)atoi(e);
    if (const char* e = getenv("MSM_AMD_SORT")) ctx->sort_atomic = (strcmp(e, "atomic") == 0);
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream_s, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return MSM_AMD_EHIP;
    }
    for (Slot& sl : ctx->slots) {
        if (hipStreamCreateWithFlags(&sl.stream_b, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&sl.stream_c, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&sl.ev_c) != hipSuccess) {
            delete ctx;
            return MSM_AMD_EHIP;
        }
        for (auto& e : sl.ev)
            if (hipEventCreate(&e) != hipSuccess) {
                delete ctx;
                return MSM_AMD_EHIP;
            }
    }
    calibrate_streams(ctx);
    *out_ctx = ctx;
    return MSM_AMD_OK;
}

void msm_amd_destroy(msm_amd_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream_s) (void)hipStreamSynchronize(ctx->stream_s);
    if (ctx->stream_s2) (void)hipStreamSynchronize(ctx->stream_s2);
    (void)hipStreamSynchronize(ctx->stream);
    for (Slot& sl : ctx->slots) {
        if (sl.stream_b) {
            (void)hipStreamSynchronize(sl.stream_b);
            (void)hipStreamDestroy(sl.stream_b);
        }
        if (sl.stream_c) {
            (void)hipStreamSynchronize(sl.stream_c);
            (void)hipStreamDestroy(sl.stream_c);
        }
        if (sl.ev_c) (void)hipEventDestroy(sl.ev_c);
        for (DevBuf* b : sl.bufs)
            if (b->p) (void)hipFree(b->p);
        if (sl.h_partials) (void)hipHostFree(sl.h_partials);
        for (auto& e : sl.ev)
            if (e) (void)hipEventDestroy(e);
    }
    if (ctx->stream_s) (void)hipStreamDestroy(ctx->stream_s);
    if (ctx->stream_s2) (void)hipStreamDestroy(ctx->stream_s2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* msm_amd_last_error(const msm_amd_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// Diagnostic: which of the context's streams serialise with each other, i.e. were mapped onto the same hardware queue (DESIGN.md
// section 7).  Streams 0..10 = A, S, B0, C0, B1, C1, B2, C2, B3, C3 and the process's null stream.  A one-wave kernel that spins for ~0.3 ms is launched on two
// streams at once: together they take ~0.3 ms on different queues, ~0.6 ms on a shared one.  out[i*11+j] = 1 when i and j serialise.
int msm_amd_debug_stream_sharing(msm_amd_ctx* ctx, uint8_t* out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st[11] = {ctx->stream, ctx->stream_s};
    st[10] = nullptr;
    for (int k = 0; k < 4; ++k) {
        st[2 + 2 * k] = ctx->slots[k].stream_b;
        st[3 + 2 * k] = ctx->slots[k].stream_c;
    }
    const uint64_t ticks = 30000;   // wall_clock64 counts at 100 MHz: 0.3 ms
    k_spin<<<1, 64, 0, st[0]>>>(1000);
    HIPCHK(ctx, hipDeviceSynchronize());
    for (int i = 0; i < 11; ++i)
        for (int j = 0; j < 11; ++j) out[i * 11 + j] = i == j;
    for (int i = 0; i < 11; ++i)
        for (int j = i + 1; j < 11; ++j) {
            auto t0 = std::chrono::steady_clock::now();
            k_spin<<<1, 64, 0, st[i]>>>(ticks);
            k_spin<<<1, 64, 0, st[j]>>>(ticks);
            HIPCHK(ctx, hipStreamSynchronize(st[i]));
            HIPCHK(ctx, hipStreamSynchronize(st[j]));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            out[i * 11 + j] = out[j * 11 + i] = us > 470.0;
        }
    return MSM_AMD_OK;
}

int msm_amd_set_option(msm_amd_ctx* ctx, const char* name, uint32_t value) {
    if (!ctx || !name) return MSM_AMD_EINVAL;
    if (!strcmp(name, "window_bits")) return msm_amd_set_window_bits(ctx, value);
    if (!strcmp(name, "segment")) { ctx->force_seg = value; return 0; }
    if (!strcmp(name, "reduce_l")) { ctx->force_L = value; return 0; }
    if (!strcmp(name, "sort_atomic")) { ctx->sort_atomic = value != 0; return 0; }
    if (!strcmp(name, "split")) { ctx->no_split = value == 0; return 0; }
    if (!strcmp(name, "glv")) { ctx->glv_mode = value > 2 ? 2 : (int)value; return 0; }
    if (!strcmp(name, "split2")) { ctx->no_split2 = value == 0; return 0; }
    if (!strcmp(name, "split2_nb")) { ctx->split2_nb = value; return 0; }
    if (!strcmp(name, "slice")) {
        if (value < 1 || value > (1u << 27)) { ctx->err = "slice must be in [1, 2^27]"; return MSM_AMD_EINVAL; }
        ctx->max_slice = value;
        return 0;
    }
    if (!strcmp(name, "slots")) {
        if (value < 1 || value > msm_amd_ctx::MAX_SLOTS) { ctx->err = "slots must be in [1, 8]"; return MSM_AMD_EINVAL; }
        for (Slot& sl : ctx->slots)
            if (sl.pending) { ctx->err = "cannot change the slot count with calls in flight"; return MSM_AMD_EINVAL; }
        ctx->nslots = value;
        ctx->next_slot = 0;
        return 0;
    }
    ctx->err = "unknown option";
    return MSM_AMD_EINVAL;
}

int msm_amd_set_window_bits(msm_amd_ctx* ctx, uint32_t c) {
    if (!ctx || c > 16) return MSM_AMD_EINVAL;
    ctx->force_c = c;
    return 0;
}

int msm_amd_multiexp(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                     uint32_t n, uint8_t* out) {
    RunArgs a{bases, scalars, scalar_size, n, flags, 0, scalar_size * 8, 0, 1, false};
    return dispatch(ctx, curve, group, a, out);
}

int msm_amd_multiexp_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                           uint32_t n, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    RunArgs a{bases, scalars, scalar_size, n, flags, 0, scalar_size * 8, 0, 1, false};
    return submit(ctx, curve, group, a, ticket);
}

int msm_amd_multiexp_shard_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars,
                                 uint32_t scalar_size, uint32_t n, uint32_t shard, uint32_t nshards, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    RunArgs a{bases, scalars, scalar_size, n, flags, 0, scalar_size * 8, shard, nshards, true};
    return submit(ctx, curve, group, a, ticket);
}

int msm_amd_wait(msm_amd_ctx* ctx, uint64_t ticket, uint8_t* out) { return finish_ticket(ctx, ticket, out); }

int msm_amd_multiexp_chunk(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars,
                           uint32_t scalar_size, uint32_t n, uint32_t start_bit, uint32_t chunk_size, uint8_t* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (chunk_size < 1 || chunk_size > 25) {   // build_multiexp.js:25-94 is exact for chunkSize <= 25
        ctx->err = "chunk_size must be in [1, 25]";
        return MSM_AMD_EINVAL;
    }
    uint32_t total = scalar_size * 8;
    uint32_t nbits = start_bit >= total ? 0u : std::min(chunk_size, total - start_bit);
    RunArgs a{bases, scalars, scalar_size, n, flags, start_bit, nbits, 0, 1, false};
    return dispatch(ctx, curve, group, a, out);
}

int msm_amd_multiexp_shard(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars,
                           uint32_t scalar_size, uint32_t n, uint32_t shard, uint32_t nshards, uint8_t* out_partial) {
    RunArgs a{bases, scalars, scalar_size, n, flags, 0, scalar_size * 8, shard, nshards, true};
    return dispatch(ctx, curve, group, a, out_partial);
}

#define BY_FIELD(curve, group, CALL)                                                         \
    do {                                                                                     \
        if ((curve) == MSM_AMD_BLS12381 && (group) == 1) { using F = FpH<Bls12381Fp>; CALL; }        \
        if ((curve) == MSM_AMD_BLS12381 && (group) == 2) { using F = Fp2T<FpH<Bls12381Fp>>; CALL; }  \
        if ((curve) == MSM_AMD_BN128 && (group) == 1) { using F = FpH<Bn254Fp>; CALL; }              \
        if ((curve) == MSM_AMD_BN128 && (group) == 2) { using F = Fp2T<FpH<Bn254Fp>>; CALL; }        \
    } while (0)

int msm_amd_combine(int curve, int group, const uint8_t* partials, uint32_t count, uint8_t* out) {
    if (!partials || !out) return MSM_AMD_EINVAL;
    BY_FIELD(curve, group, return combine_t<F>(partials, count, out));
    return MSM_AMD_EINVAL;
}

int msm_amd_normalize(int curve, int group, const uint8_t* jac_in, uint8_t* jac_out) {
    if (!jac_in || !jac_out) return MSM_AMD_EINVAL;
    BY_FIELD(curve, group, {
        host_store_jacobian_normalised(host_load_jacobian<F>(jac_in), jac_out);
        return 0;
    });
    return MSM_AMD_EINVAL;
}
int msm_amd_to_affine(int curve, int group, const uint8_t* jac_in, uint8_t* aff_out) {
    if (!jac_in || !aff_out) return MSM_AMD_EINVAL;
    BY_FIELD(curve, group, {
        Affine<F> a = XYZZ<F>::to_affine(host_load_jacobian<F>(jac_in));
        memcpy(aff_out, &a, sizeof(a));
        return 0;
    });
    return MSM_AMD_EINVAL;
}
int msm_amd_eq(int curve, int group, const uint8_t* jac_a, const uint8_t* jac_b) {
    if (!jac_a || !jac_b) return MSM_AMD_EINVAL;
    BY_FIELD(curve, group, {
        XYZZ<F> d = XYZZ<F>::add(host_load_jacobian<F>(jac_a), XYZZ<F>::neg(host_load_jacobian<F>(jac_b)));
        return d.is_inf() ? 1 : 0;
    });
    return MSM_AMD_EINVAL;
}
int msm_amd_f_to_montgomery(int curve, const uint8_t* in, uint32_t count, uint8_t* out) {
    if (!in || !out) return MSM_AMD_EINVAL;
    if (curve == MSM_AMD_BLS12381) return f_conv<FpH<Bls12381Fp>, true>(in, count, out);
    if (curve == MSM_AMD_BN128) return f_conv<FpH<Bn254Fp>, true>(in, count, out);
    return MSM_AMD_EINVAL;
}
int msm_amd_f_from_montgomery(int curve, const uint8_t* in, uint32_t count, uint8_t* out) {
    if (!in || !out) return MSM_AMD_EINVAL;
    if (curve == MSM_AMD_BLS12381) return f_conv<FpH<Bls12381Fp>, false>(in, count, out);
    if (curve == MSM_AMD_BN128) return f_conv<FpH<Bn254Fp>, false>(in, count, out);
    return MSM_AMD_EINVAL;
}

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

int msm_amd_fft(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* in, uint32_t n, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out) {
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    if ((n & (n - 1)) != 0 || n > (1u << 27)) {
        ctx->err = "fft: n must be a power of two, at most 2^27";
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
    ctx->err = "unknown curve/group";
    return MSM_AMD_EINVAL;
}
int msm_amd_batch_apply_key(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* in, uint32_t n, const uint8_t* first32,
                            const uint8_t* inc32, void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out || !first32 || !inc32) {
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    if (curve == MSM_AMD_BLS12381 && group == 1) return apply_key_t<Fp<Bls12381Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BLS12381 && group == 2) return apply_key_t<Fp2<Bls12381Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BN128 && group == 1) return apply_key_t<Fp<Bn254Fp>>(ctx, flags, in, n, first32, inc32, out);
    if (curve == MSM_AMD_BN128 && group == 2) return apply_key_t<Fp2<Bn254Fp>>(ctx, flags, in, n, first32, inc32, out);
    ctx->err = "unknown curve/group";
    return MSM_AMD_EINVAL;
}
int msm_amd_fr_batch_apply_key(msm_amd_ctx* ctx, int curve, uint32_t flags, const void* in, uint32_t n, const uint8_t* first32, const uint8_t* inc32,
                               void* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (!in || !out || !first32 || !inc32 || (curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128)) {
        ctx->err = "bad argument";
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
            ctx->err = "hipMalloc failed";
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
        ctx->err = std::string("fr_batch_apply_key: ") + hipGetErrorString(e);
        return MSM_AMD_EHIP;
    }
    return 0;
}

int msm_amd_batch_in_group_affine(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* points, uint32_t n, uint8_t* out_flags) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (n && (!points || !out_flags)) {
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    if (curve != MSM_AMD_BLS12381 || (group != 1 && group != 2)) {
        ctx->err = "inGroupAffine exists for BLS12-381 only (as in the reference)";
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
        ctx->err = "null pointer";
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
    ctx->err = "unknown curve/group";
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
        ctx->err = "null pointer";
        return MSM_AMD_EINVAL;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t pb = (size_t)n * 96, sb = (size_t)n * 32;
    if (flags & MSM_AMD_DEVICE_PTRS) {
        if (((uintptr_t)points & 15) || ((uintptr_t)points_out & 15) || ((uintptr_t)scalars & 3) || ((uintptr_t)scalars_out & 3)) {
            ctx->err = "device pointers must be 16-byte (points) / 4-byte (scalars) aligned";
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
        ctx->err = std::string("glv_preprocess: ") + hipGetErrorString(e);
        return MSM_AMD_EHIP;
    }
    return 0;
}

int msm_amd_dev_alloc(msm_amd_ctx* ctx, size_t bytes, void** dev_ptr) {
    if (!ctx || !dev_ptr) return MSM_AMD_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMalloc(dev_ptr, bytes ? bytes : 16));
    return 0;
}
int msm_amd_dev_free(msm_amd_ctx* ctx, void* dev_ptr) {
    if (!ctx) return MSM_AMD_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipFree(dev_ptr));
    return 0;
}
int msm_amd_h2d(msm_amd_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
    if (!ctx) return MSM_AMD_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return 0;
}
int msm_amd_d2h(msm_amd_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!ctx) return MSM_AMD_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int msm_amd_gen_points(msm_amd_ctx* ctx, int curve, int group, uint64_t seed, uint32_t n, void* dev_out) {
    if (!ctx || !dev_out) return MSM_AMD_EINVAL;
    if (n == 0) return 0;
    if (curve == MSM_AMD_BLS12381 && group == 1) {
        using P = Bls12381Fp;
        Affine<Fp<P>> g{fp_from<P>(P::G1X), fp_from<P>(P::G1Y)};
        return gen_points_t<Fp<P>>(ctx, g, seed, n, dev_out);
    }
    if (curve == MSM_AMD_BN128 && group == 1) {
        using P = Bn254Fp;
        Affine<Fp<P>> g{fp_from<P>(P::G1X), fp_from<P>(P::G1Y)};
        return gen_points_t<Fp<P>>(ctx, g, seed, n, dev_out);
    }
    if (curve == MSM_AMD_BLS12381 && group == 2) {
        using P = Bls12381Fp;
        Affine<Fp2<P>> g{{fp_from<P>(P::G2X0), fp_from<P>(P::G2X1)}, {fp_from<P>(P::G2Y0), fp_from<P>(P::G2Y1)}};
        return gen_points_t<Fp2<P>>(ctx, g, seed, n, dev_out);
    }
    if (curve == MSM_AMD_BN128 && group == 2) {
        using P = Bn254Fp;
        Affine<Fp2<P>> g{{fp_from<P>(P::G2X0), fp_from<P>(P::G2X1)}, {fp_from<P>(P::G2Y0), fp_from<P>(P::G2Y1)}};
        return gen_points_t<Fp2<P>>(ctx, g, seed, n, dev_out);
    }
    ctx->err = "unknown curve/group";
    return MSM_AMD_EINVAL;
}

int msm_amd_get_timings(const msm_amd_ctx* ctx, float* ms, int capacity) {
    if (!ctx || !ms) return MSM_AMD_EINVAL;
    int k = std::min(capacity, 8);
    for (int i = 0; i < k; ++i) ms[i] = ctx->timings[i];
    return k;
}
int msm_amd_get_geometry(const msm_amd_ctx* ctx, uint32_t* out6) {
    if (!ctx || !out6) return MSM_AMD_EINVAL;
    for (int i = 0; i < 6; ++i) out6[i] = ctx->geometry[i];
    return 0;
}

#define MSM_AMD_DEF(curve, g, CID, GID)                                                                                                    \
    int msm_##curve##_##g##_multiexp(msm_amd_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint32_t scalar_size, uint32_t n,     \
                                     uint8_t* out) {                                                                                       \
        return msm_amd_multiexp(ctx, CID, GID, MSM_AMD_BASES_JACOBIAN, bases, scalars, scalar_size, n, out);                               \
    }                                                                                                                                      \
    int msm_##curve##_##g##_multiexpAffine(msm_amd_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint32_t scalar_size,           \
                                           uint32_t n, uint8_t* out) {                                                                     \
        return msm_amd_multiexp(ctx, CID, GID, 0, bases, scalars, scalar_size, n, out);                                                    \
    }                                                                                                                                      \
    int msm_##curve##_##g##_multiexp_chunk(msm_amd_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint32_t scalar_size,           \
                                           uint32_t n, uint32_t start_bit, uint32_t chunk_size, uint8_t* out) {                            \
        return msm_amd_multiexp_chunk(ctx, CID, GID, MSM_AMD_BASES_JACOBIAN, bases, scalars, scalar_size, n, start_bit, chunk_size, out);  \
    }                                                                                                                                      \
    int msm_##curve##_##g##_multiexpAffine_chunk(msm_amd_ctx* ctx, const uint8_t* bases, const uint8_t* scalars, uint32_t scalar_size,     \
                                                 uint32_t n, uint32_t start_bit, uint32_t chunk_size, uint8_t* out) {                      \
        return msm_amd_multiexp_chunk(ctx, CID, GID, 0, bases, scalars, scalar_size, n, start_bit, chunk_size, out);                       \
    }
MSM_AMD_DEF(bls12381, g1m, MSM_AMD_BLS12381, 1)
MSM_AMD_DEF(bls12381, g2m, MSM_AMD_BLS12381, 2)
MSM_AMD_DEF(bn128, g1m, MSM_AMD_BN128, 1)
MSM_AMD_DEF(bn128, g2m, MSM_AMD_BN128, 2)

}  // extern "C"
