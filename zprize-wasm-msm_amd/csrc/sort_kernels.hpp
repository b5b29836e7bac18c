// sort_kernels.hpp -- stages 1-3 of the engine (scalar windowing = getChunk, build_multiexp.js:25-94, and
// the radix-c bucket scatter that replaces the reference's table walk :186-233): digit extraction with signed recoding, the two-pass
// LDS radix partition by bucket, the scans and the ordering of buckets by population.  HBM-bound kernels; included by engine.hip only.
#pragma once
#include "engine_common.hpp"
#include "glv.hpp"

namespace msm {

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

}  // namespace msm
