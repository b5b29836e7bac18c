// sort_kernels.hpp -- stages 1-3 of the engine (scalar windowing = getChunk, build_multiexp.js:25-94, and
// the radix-c bucket scatter that replaces the reference's table walk :186-233): signed-digit recoding of the scalars and a
// two-level LDS radix partition of the (pair, window) entries by bucket.  HBM-bound kernels; included by engine.hip only.
//
//   k_glv_halves      (GLV schedule only) scalar -> two signed 127-bit halves, written once as 16-byte items
//   k_tile_hist       one workgroup per TILE of scalars, all windows: digits are recomputed from the scalar held in registers
//                     (never stored), LDS histogram over the COARSE bits of the bucket index, one global atomic per non-empty
//                     (window, bin) -> bin populations;  k_bin_scan: their exclusive scan = bin starts and cursors
//   k_tile_scatter    same tiles, same digits: a counting sort of the tile's entries of four windows at a time in LDS, every
//                     (tile, window, bin) run reserves its place with one atomic on the bin's cursor and is stored by consecutive
//                     lanes -> partition entries {fine key | sign | index}, 4 bytes at every size: an index wider than its field
//                     puts its high bits into the bin (index slices, SortGeom)
//   k_fine_small      one workgroup per (window, coarse bin) whose span fits the registers of a workgroup (<= FINE_CAP entries:
//                     every call up to n ~ 2^20): the span is read ONCE, bucket count/offset come from an LDS histogram, the
//                     span is permuted in LDS and written back as one coalesced stream
//   k_big_*           longer spans (large n, the dense bins of a partly used top window, adversarial scalars): cut into chunks
//                     that workgroups sort in parallel (histogram pass, per-bin offsets, scatter pass)
//   k_order_*         buckets ordered by population so that the 64 lanes of an accumulation wave run equal trip counts
// Global atomics only on a few thousand bin / bucket cursors (one per run, never per entry); every data access is a coalesced
// stream or a run inside one bin.  The order of the entries inside a bucket is not deterministic -- the sums do not depend on it.
//
// Round 1 kept 16-bit digit codes in HBM (written once, read twice), 8-byte entries (written once, read twice) and a (tile x bin)
// count matrix with its scan: 713 MB per call at n = 2^20 against 101 MB algorithmic.  Recomputing the digits and reading each
// span once brings that to ~270 MB, and nothing is 16 bits wide any more: windows up to 21 bits (8 coarse + 12 fine bits).
// Round 2 still switched to 8-byte entries beyond n = 2^23; round 3 keeps 4 bytes there too (the template parameter E survives
// with its one instantiation, uint32_t).
#pragma once
#include "engine_common.hpp"
#include "glv.hpp"

namespace msm {

constexpr int SORT_BLOCK = 256;
constexpr int COARSE_MAX = 512;        // coarse bins per window (9 bits: round 4 -- 8 until then)
constexpr int FINE_MAX_BITS = 12;      // fine buckets per coarse bin (4096)
static_assert(MAX_WINDOW_BITS <= 1 + 9 + FINE_MAX_BITS && MAX_WINDOW_BITS - 1 + 12 <= 32, "sign + coarse + fine bits of a window; bucket index and a 12-bit rank share a word (k_tile_scatter)");

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

// ------------------------------------------------------------------------------------------
// Digit source of one tile: ITEMS items per lane, consumed window by window from the low end.
//   NW = 8: whole 32-byte scalars held in registers (16-byte aligned input), shifted right by c after every window
//   NW = 4: 16-byte GLV halves (bit 127 = sign of the half), same
//   NW = 0: any scalar size / bit range: the bits of every window are read from memory (L1/L2-resident tile)
// next() returns the signed digit of the current window: magnitude in [0, 2^(c-1)] and whether it is negative
// (digit in [-2^(c-1), 2^(c-1)], carry into the next window).
// ------------------------------------------------------------------------------------------
template <int NW, int ITEMS, int BLOCK>
struct DigitSource {
    uint32_t w[ITEMS][NW > 0 ? NW : 1];
    uint32_t carry;   // bit k: carry of item k
    uint32_t flip;    // bit k: the item's own sign (GLV halves)
    uint32_t valid;   // bit k: item exists
    uint32_t first;   // index of this lane's item 0 (items are first + k * BLOCK)

    MSM_D void load(const uint8_t* __restrict__ scalars, const Geom& g, uint32_t tile_base) {
        carry = flip = valid = 0;
        first = tile_base + threadIdx.x;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = first + k * BLOCK;
            if (i < g.n) valid |= 1u << k;
            if constexpr (NW > 0) {
                if (i < g.n) {
                    const uint4* sp = reinterpret_cast<const uint4*>(scalars) + (size_t)i * (NW / 4);
#pragma unroll
                    for (int q = 0; q < NW / 4; ++q) {
                        uint4 v = sp[q];
                        w[k][4 * q] = v.x;
                        w[k][4 * q + 1] = v.y;
                        w[k][4 * q + 2] = v.z;
                        w[k][4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NW; ++q) w[k][q] = 0;
                }
                if constexpr (NW == 4) {
                    if (w[k][3] >> 31) flip |= 1u << k;
                    w[k][3] &= 0x7fffffffu;
                }
            }
        }
    }
    // digit of item k in window `win` (windows must be visited in order 0, 1, 2, ...).  Windows below g.ns are c bits wide and
    // signed; the rest (prepared bases only) c - 1 bits wide and unsigned: they absorb a carry, never produce one.
    MSM_D void next(const uint8_t* __restrict__ scalars, const Geom& g, int k, uint32_t win, uint32_t& mag, bool& neg) {
        const bool sgn = win < g.ns;
        const uint32_t wb = sgn ? g.c : g.c - 1u;
        uint32_t raw;
        if constexpr (NW > 0) {
            raw = w[k][0] & ((1u << wb) - 1u);   // bits beyond the scalar are zero by construction
#pragma unroll
            for (int q = 0; q < NW - 1; ++q) w[k][q] = __builtin_amdgcn_alignbit(w[k][q + 1], w[k][q], wb);
            w[k][NW - 1] >>= wb;
        } else {
            const uint32_t off = window_start_bit(g, win);
            const uint32_t nb = off >= g.nbits ? 0u : min(wb, g.nbits - off);
            raw = (valid >> k) & 1u ? scalar_bits(scalars, g, first + k * BLOCK, g.bit_lo + off, nb) : 0u;
        }
        raw += (carry >> k) & 1u;
        const bool negd = sgn && raw > g.half;
        mag = negd ? (1u << g.c) - raw : raw;
        carry = negd ? (carry | (1u << k)) : (carry & ~(1u << k));
        neg = negd != (((flip >> k) & 1u) != 0);
    }
    // After the last window: does item k hold scalar bits beyond the windows?  (Prepared bases cover the scalar width they were
    // prepared for; a wider scalar cannot be served by the table and is reported, never silently truncated.)
    MSM_D bool leftover(const uint8_t* __restrict__ scalars, const Geom& g, int k) const {
        if constexpr (NW > 0) {
            uint32_t any = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q) any |= w[k][q];
            return any != 0;
        } else {
            if (!((valid >> k) & 1u)) return false;
            uint32_t any = 0;
            for (uint32_t off = window_start_bit(g, g.W); off < g.nbits; off += 24u)
                any |= scalar_bits(scalars, g, first + k * BLOCK, g.bit_lo + off, min(24u, g.nbits - off));
            return any != 0;
        }
    }
};

// The few words the host wants beside the window partials -- where the last bucket's entries end (= the number of entries), the tail
// counters, the too-wide flag of a prepared table, where the top owned window's entries start -- gathered behind the partials so that
// ONE copy brings everything back (six separate 4- to 16-byte copies cost ~40 us at the end of every call).
// Having read them it zeroes the four tail counters and the two flag words: they are all zero between the calls of a slot (round 4 -- the
// two 16-byte memsets at the head of every call's sort and accumulation are gone; Slot::ctl_zero / flags_zero say when that holds).
__global__ void k_collect_tail(const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count, uint32_t NB, uint32_t top_first,
                               uint32_t* __restrict__ tailctl, uint32_t* __restrict__ flags, uint32_t* __restrict__ out) {
    if (threadIdx.x == 0) {
        out[0] = offset[NB - 1];
        out[1] = count[NB - 1];
        out[2] = tailctl[0];
        out[3] = tailctl[1];
        out[4] = tailctl[2];
        out[5] = tailctl[3];
        out[6] = flags[0];
        out[7] = offset[top_first];
        out[8] = flags[1];   // significant bits of the widest scalar (k_tile_hist; 0 = not measured)
        tailctl[0] = tailctl[1] = tailctl[2] = tailctl[3] = 0;
        flags[0] = flags[1] = flags[2] = flags[3] = 0;
    }
}

// GLV schedule (BLS12-381 G1; SURVEY.md 8f rank 1, constants of build_glv.js:15-22): every scalar is split into two signed
// 127-bit halves (glv::split_centered).  Virtual item 2i is (P_i, k1), item 2i+1 is (phi(P_i), k2) with phi(x, y) = (beta*x, y):
// 2n items of 128 bits, 8 windows of 16 bits and no carry out of the last one.  The halves are written once (|k| in bits
// 0..126, the sign in bit 127) and the tile kernels read them like 16-byte scalars.
__global__ void __launch_bounds__(256) k_glv_halves(const uint8_t* __restrict__ scalars, uint32_t n, uint4* __restrict__ halves) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars) + (size_t)i * 2;
    uint4 lo = sp[0], hi = sp[1];
    uint32_t k[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t h[2][4];
    uint32_t sg = glv::split_centered(k, h[0], h[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) halves[(size_t)2 * i + j] = make_uint4(h[j][0], h[j][1], h[j][2], h[j][3] | (((sg >> j) & 1u) << 31));
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

// Layout of the partition: bucket index (0-based magnitude - 1, c-1 bits) = coarse bin (top cbits) || fine key (low fbits).
struct SortGeom {
    uint32_t cbits, fbits;   // c - 1 = cbits + fbits; cbits <= 8, fbits <= 12
    uint32_t cb;             // coarse bins per window = 2^cbits
    uint32_t nbins;          // owned windows x coarse bins ("parent" bins: one span of the fine sort each)
    uint32_t idxbits;        // bits of the index field of a 4-byte entry {fine key | sign | index}: fine key above the sign bit at position idxbits
    // Indices wider than idxbits (n > 2^23 at c = 16, > 2^20 at c = 20; the W x n table indices of prepared bases): every parent bin
    // is cut into S sub-bins by the HIGH bits of the index -- sub-bin s holds the entries with index >> idxbits == s, in the order
    // s = 0, 1, ... inside the parent's span -- and an entry keeps the low idxbits only.  The position of an entry in its parent's
    // span says which sub-bin it is in, so the fine sort puts the high bits back.  A tile of scalars lies in ONE slice of the index
    // space (2^idxbits is a multiple of the tile size), so the partition kernels know s per (tile, window) and keep one reservation
    // per run.  S = 1: plain entries, as before.  Entries stay 4 bytes at every size (round 2 switched to 8 bytes there: twice the
    // partition traffic, three passes over it).
    uint32_t S;
};
MSM_D uint32_t slice_of_index(const SortGeom& sg, uint32_t index) { return sg.S > 1 ? index >> sg.idxbits : 0u; }
// sub-bin of the entry at position p of a parent span whose S + 1 sub-bin boundaries are bnd[0..S] (non-decreasing, absolute positions;
// empty sub-bins repeat a value): the largest k < S with bnd[k] <= p
MSM_D uint32_t slice_of_position(const uint32_t* bnd, uint32_t S, uint32_t p) {
    uint32_t lo = 0, hi = S;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bnd[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------
// k_tile_hist: binhist[wl * cb + bin] += entries of this tile that fall into (owned window wl, coarse bin)
// (LDS histogram per tile, then one global atomic per non-empty (window, bin): ~2000 counters shared by all tiles)
// ------------------------------------------------------------------------------------------
template <int NW, int ITEMS, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_tile_hist(const uint8_t* __restrict__ scalars, Geom g, SortGeom sg, uint32_t* __restrict__ binhist,
                                                      uint32_t* __restrict__ too_wide) {
    extern __shared__ uint32_t smem_u32[];
    // [rows][cb]: one row per owned window -- per window of a prepared table, whose windows share the parent bins but not the slice
    // of the index space their entries fall into
    uint32_t* h = smem_u32;
    const uint32_t tile = blockIdx.x;
    const uint32_t rows = g.merged ? g.W : g.Wl, nh = rows * sg.cb;
    __shared__ uint32_t widest;   // of this workgroup's scalars
    // (read at the top, used after the last barrier: the round trip is hidden behind the pass)
    const uint32_t widest_so_far = NW == 8 && threadIdx.x == 0 ? *reinterpret_cast<const volatile uint32_t*>(too_wide + 1) : 0u;
    if (threadIdx.x == 0) widest = 0;
    for (uint32_t s = threadIdx.x; s < nh; s += BLOCK) h[s] = 0;
    __syncthreads();
    DigitSource<NW, ITEMS, BLOCK> ds;
    ds.load(scalars, g, tile * (ITEMS * BLOCK));
    if constexpr (NW == 8) {
        // the widest scalar of the call (significant bits), for the planner of the context's NEXT calls: scalars much narrower than their
        // 32 bytes -- 128-bit challenges, 64-bit witness values -- are better served by windows that tile the width they use (engine_call.hpp)
        uint32_t mb = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            uint32_t top = 0;
#pragma unroll
            for (int q = 0; q < NW; ++q)
                if (ds.w[k][q]) top = 32u * q + 32u - (uint32_t)__clz(ds.w[k][q]);
            mb = max(mb, top);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, d, 64));
        // (one global atomic per wave on ONE address serialises -- 4096 of them took 30 us of this kernel's 55 at n = 2^20: the waves meet
        // in LDS, and only a workgroup that would raise the maximum sends one, at its end; uniform scalars settle it within the first few)
        if ((threadIdx.x & 63u) == 0 && mb) atomicMax(&widest, mb);
    }
    for (uint32_t w = 0; w < g.W; ++w) {
        const bool owned = g.merged || (w % g.nshards) == g.shard;
        const uint32_t wl = g.merged ? w : w / g.nshards;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            uint32_t mag;
            bool neg;
            ds.next(scalars, g, k, w, mag, neg);
            if (owned && mag != 0) atomicAdd(&h[wl * sg.cb + ((mag - 1) >> sg.fbits)], 1u);
        }
    }
    if (g.check_wide) {   // (uniform) scalar bits the windows do not cover: a prepared table's, or the narrow layout's
        bool wide = false;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) wide |= ds.leftover(scalars, g, k);
        if (wide) atomicOr(too_wide, 1u);
    }
    __syncthreads();
    if (NW == 8 && threadIdx.x == 0 && widest > widest_so_far) atomicMax(too_wide + 1, widest);
    for (uint32_t s = threadIdx.x; s < nh; s += BLOCK) {
        const uint32_t v = h[s];
        if (v) {
            const uint32_t row = s / sg.cb, bin = s % sg.cb;
            const uint32_t parent = g.merged ? bin : row * sg.cb + bin;
            const uint32_t slice = slice_of_index(sg, (g.merged ? row * g.stride : 0u) + tile * (ITEMS * BLOCK));
            // (slice-major: the tiles of one slice share nbins consecutive counters -- a few dozen cache lines, as without slices;
            // parent-major put every counter of a tile on a line of its own and the pass took 3.5 times as long at n = 2^26)
            atomicAdd(&binhist[slice * sg.nbins + parent], v);
        }
    }
}

// single workgroup: the exclusive scan of the bin populations.  Positions run over the bins in parent-major order -- all sub-bins
// (index slices) of a parent bin one after the other, so that a parent's span is contiguous: binstart[parent * S + slice], with
// binstart[nparent * S] = total -- while the counters the tiles add to (binhist) and reserve from (cursor) are kept slice-major,
// [slice][parent]: the tiles of one slice then share a few dozen cache lines.  Leaves binhist zeroed for the slot's next call and
// zeroes the key counters of the population ordering (two fill launches less).
__global__ void __launch_bounds__(1024) k_bin_scan(uint32_t* __restrict__ binhist, uint32_t nparent, uint32_t S, uint32_t* __restrict__ binstart,
                                                    uint32_t* __restrict__ cursor, uint32_t* __restrict__ order_bins, uint32_t n_order_bins) {
    __shared__ uint32_t wsum[16];
    const uint32_t t = threadIdx.x, nbins = nparent * S;
    // lane t owns the `per` consecutive (parent-major) bins from t * per: their sum, a block-wide exclusive scan of the 1024 sums, then
    // the positions.  (One pass of 1024 bins per iteration with two barriers each took 0.47 ms for the 213 000 bins of n = 2^26.)
    const uint32_t per = (nbins + 1023u) / 1024u, lo = min(t * per, nbins), hi = min(lo + per, nbins);
    auto src_of = [&](uint32_t i) { return S > 1 ? (i % S) * nparent + i / S : i; };   // the same bin, slice-major
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += binhist[src_of(i)];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(t & 63u) >= d) incl += o;
    }
    if ((t & 63u) == 63u) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        if (q < (t >> 6)) run += wsum[q];
        total += wsum[q];
    }
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t src = src_of(i), v = binhist[src];
        binstart[i] = run;
        cursor[src] = run;
        binhist[src] = 0;
        run += v;
    }
    if (t == 0) binstart[nbins] = total;
    for (uint32_t i = t; i < n_order_bins; i += 1024) order_bins[i] = 0;
}

// The same scan by SEVERAL workgroups, for the bin counts of large calls (213 000 bins at n = 2^26: the single workgroup took 0.32 ms):
// pass A leaves the population of every workgroup's contiguous (parent-major) range of bins in blocksum[], pass B starts each range at
// the sum of the ranges before it.  gridDim.x <= 64.
__global__ void __launch_bounds__(1024) k_bin_scan_sums(const uint32_t* __restrict__ binhist, uint32_t nparent, uint32_t S, uint32_t* __restrict__ blocksum) {
    __shared__ uint32_t wsum[16];
    const uint32_t t = threadIdx.x, nbins = nparent * S;
    const uint32_t range = (nbins + gridDim.x - 1) / gridDim.x, r0 = min(blockIdx.x * range, nbins), r1 = min(r0 + range, nbins);
    auto src_of = [&](uint32_t i) { return S > 1 ? (i % S) * nparent + i / S : i; };
    uint32_t sum = 0;
    for (uint32_t i = r0 + t; i < r1; i += 1024) sum += binhist[src_of(i)];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((t & 63u) == 0) wsum[t >> 6] = sum;
    __syncthreads();
    if (t == 0) {
        uint32_t tot = 0;
        for (int q = 0; q < 16; ++q) tot += wsum[q];
        blocksum[blockIdx.x] = tot;
    }
}
__global__ void __launch_bounds__(1024) k_bin_scan_ranges(uint32_t* __restrict__ binhist, uint32_t nparent, uint32_t S, const uint32_t* __restrict__ blocksum,
                                                           uint32_t* __restrict__ binstart, uint32_t* __restrict__ cursor, uint32_t* __restrict__ order_bins,
                                                           uint32_t n_order_bins) {
    __shared__ uint32_t wsum[16];
    const uint32_t t = threadIdx.x, nbins = nparent * S;
    const uint32_t range = (nbins + gridDim.x - 1) / gridDim.x, r0 = min(blockIdx.x * range, nbins), r1 = min(r0 + range, nbins);
    uint32_t base = 0, grand = 0;
    for (uint32_t w = 0; w < gridDim.x; ++w) {   // (at most 64 values, the same for every lane: scalar loads)
        const uint32_t v = blocksum[w];
        if (w < blockIdx.x) base += v;
        grand += v;
    }
    const uint32_t per = (r1 - r0 + 1023u) / 1024u, lo = min(r0 + t * per, r1), hi = min(lo + per, r1);
    auto src_of = [&](uint32_t i) { return S > 1 ? (i % S) * nparent + i / S : i; };
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += binhist[src_of(i)];
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(t & 63u) >= d) incl += o;
    }
    if ((t & 63u) == 63u) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t run = base + incl - sum;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q)
        if (q < (t >> 6)) run += wsum[q];
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t src = src_of(i), v = binhist[src];
        binstart[i] = run;
        cursor[src] = run;
        binhist[src] = 0;
        run += v;
    }
    if (blockIdx.x == 0) {
        if (t == 0) binstart[nbins] = grand;
        for (uint32_t i = t; i < n_order_bins; i += 1024) order_bins[i] = 0;
    }
}

// partition entry
template <class E>
MSM_D E make_entry(uint32_t idx, bool neg, uint32_t fine, uint32_t idxbits);
template <>
MSM_D uint32_t make_entry<uint32_t>(uint32_t idx, bool neg, uint32_t fine, uint32_t idxbits) {
    return idx | ((neg ? 1u : 0u) << idxbits) | (fine << (idxbits + 1));
}
MSM_D uint32_t entry_fine(uint32_t e, uint32_t idxbits) { return e >> (idxbits + 1); }
// index with the sign in bit 31: what the accumulation kernels read
MSM_D uint32_t entry_sorted(uint32_t e, uint32_t idxbits) { return (e & ((1u << idxbits) - 1u)) | (((e >> idxbits) & 1u) << 31); }

// block-wide exclusive scan of nf counters in LDS, in place; returns their sum.  Lane t owns the consecutive counters
// [t*per, (t+1)*per), per = ceil(nf / BLOCK).  wsum: BLOCK / 64 words of LDS.
template <int BLOCK = SORT_BLOCK>
MSM_D uint32_t block_exclusive_scan(uint32_t* a, uint32_t nf, uint32_t* wsum) {
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nf + BLOCK - 1) / BLOCK;
    uint32_t s = 0;
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t i = t * per + j;
        if (i < nf) s += a[i];
    }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(t & 63u) >= d) incl += o;
    }
    if ((t & 63u) == 63u) wsum[t >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - s, total = 0;
#pragma unroll
    for (int q = 0; q < BLOCK / 64; ++q) {
        if (q < (int)(t >> 6)) run += wsum[q];
        total += wsum[q];
    }
    for (uint32_t j = 0; j < per; ++j) {
        const uint32_t i = t * per + j;
        if (i < nf) {
            const uint32_t v = a[i];
            a[i] = run;
            run += v;
        }
    }
    __syncthreads();
    return total;
}

// ------------------------------------------------------------------------------------------
// k_tile_scatter: the tile's entries of GW windows at a time are counting-sorted by (window, coarse bin) in LDS -- the rank of an
// entry inside its run comes back from the LDS histogram's atomic -- every non-empty run reserves its place in the bin with ONE
// global atomic on the bin's cursor, and the staged runs are stored by consecutive lanes (a run of a 4096-item tile is ~32 entries,
// one 128-byte line).  Storing every entry straight from its lane instead (no staging) was measured: 64 separate 4-byte requests
// per wave instruction, 90 us against 50 for the whole kernel (profiles/r02_sort_variants.txt).  The order of the tiles inside a
// bin is whatever order the reservations arrive in: sums do not care.
// ------------------------------------------------------------------------------------------
template <int NW, int ITEMS, int BLOCK, class E, int GW>
__global__ void __launch_bounds__(BLOCK) k_tile_scatter(const uint8_t* __restrict__ scalars, Geom g, SortGeom sg, uint32_t* __restrict__ cursor,
                                                         E* __restrict__ tmp) {
    constexpr int T = ITEMS * BLOCK;
    // (bucket index << RB) | rank inside the run: 12 bits of rank for tiles up to 4096 items (bucket indices up to 20 bits: c <= 21),
    // 13 for the 8192-item tiles of the 512-bin partition (bucket indices up to 19 bits: c <= 20 -- the engine only pairs them so)
    constexpr uint32_t RB = T <= 4096 ? 12u : 13u;
    static_assert(T <= 8192, "ranks are packed into 12 or 13 bits");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    E* stage = reinterpret_cast<E*>(smem_raw);                                  // [GW * T] entries grouped by run
    uint32_t* hs = reinterpret_cast<uint32_t*>(stage + GW * T);                 // [GW * cb] counts, then the start of each run in `stage`
    uint32_t* gb = hs + GW * COARSE_MAX;                                        // [GW * cb] where the run goes in tmp
    uint32_t* wsum = gb + GW * COARSE_MAX;                                      // [BLOCK / 64]
    uint16_t* srun = reinterpret_cast<uint16_t*>(wsum + BLOCK / 64);            // [GW * T] run of every staged entry
    const uint32_t tile = blockIdx.x, t = threadIdx.x;
    DigitSource<NW, ITEMS, BLOCK> ds;
    ds.load(scalars, g, tile * T);
    const uint32_t ncnt = GW * sg.cb;
    for (uint32_t w0 = 0; w0 < g.W; w0 += GW) {
        // (bucket index = magnitude - 1, < 2^20) << RB | rank inside the run (< T).  Validity is kept apart (livem): with the
        // magnitude itself packed, a digit of exactly 2^20 (c = 21) shifted out of the word -- the entry was dropped or indexed LDS
        // out of bounds -- and every value of the packed word is a legitimate (bucket, rank) pair, so none is free as a marker.
        uint32_t mr[GW][ITEMS];
        uint32_t negm[GW], livem[GW];
        bool any = false;
        for (uint32_t s = t; s < ncnt; s += BLOCK) hs[s] = 0;
        __syncthreads();
#pragma unroll
        for (int gw = 0; gw < GW; ++gw) {
            const uint32_t w = w0 + gw;
            const bool owned = w < g.W && (g.merged || (w % g.nshards) == g.shard);
            any |= owned;
            negm[gw] = livem[gw] = 0;
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                mr[gw][k] = 0;
                if (w < g.W) {
                    uint32_t mag;
                    bool neg;
                    ds.next(scalars, g, k, w, mag, neg);
                    if (neg) negm[gw] |= 1u << k;
                    if (owned && mag != 0) {
                        livem[gw] |= 1u << k;
                        mr[gw][k] = ((mag - 1) << RB) | atomicAdd(&hs[gw * sg.cb + ((mag - 1) >> sg.fbits)], 1u);
                    }
                }
            }
        }
        __syncthreads();
        if (!any) continue;   // uniform over the workgroup (the next iteration's clearing is behind its own barrier)
        // every non-empty run reserves its span of the bin
        for (uint32_t s = t; s < ncnt; s += BLOCK) {
            const uint32_t w = w0 + s / sg.cb, bin = s % sg.cb, cnt = hs[s];
            // (cnt != 0 only for owned windows; the run's sub-bin: the slice of the index space this tile's entries of window w lie in)
            const uint32_t parent = (g.merged ? 0u : w / g.nshards) * sg.cb + bin;
            gb[s] = cnt ? atomicAdd(&cursor[slice_of_index(sg, (g.merged ? w * g.stride : 0u) + tile * T) * sg.nbins + parent], cnt) : 0u;   // (slice-major, like binhist)
        }
        const uint32_t staged = block_exclusive_scan<BLOCK>(hs, ncnt, wsum);   // hs: start of each run inside the staged group
#pragma unroll
        for (int gw = 0; gw < GW; ++gw) {
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                if ((livem[gw] >> k) & 1u) {
                    const uint32_t b = mr[gw][k] >> RB, run = gw * sg.cb + (b >> sg.fbits), slot = hs[run] + (mr[gw][k] & ((1u << RB) - 1u));
                    // (prepared bases: the entry addresses the table point of this window, w * stride + i)
                    const uint32_t index = (g.merged ? (w0 + gw) * g.stride : 0u) + ds.first + k * BLOCK;
                    stage[slot] = make_entry<E>(sg.S > 1 ? index & ((1u << sg.idxbits) - 1u) : index, (negm[gw] >> k) & 1u, b & ((1u << sg.fbits) - 1u), sg.idxbits);
                    srun[slot] = (uint16_t)run;
                }
            }
        }
        __syncthreads();
        for (uint32_t s = t; s < staged; s += BLOCK) {
            const uint32_t run = srun[s];
            tmp[gb[run] + (s - hs[run])] = stage[s];
        }
        __syncthreads();
    }
}
// dynamic LDS of k_tile_scatter
template <class E>
constexpr size_t tile_scatter_lds(int T, int GW, int BLOCK) {
    return (size_t)GW * T * sizeof(E) + (size_t)2 * GW * COARSE_MAX * 4 + (size_t)(BLOCK / 64) * 4 + (size_t)GW * T * 2;
}

// Ordering of buckets by population (descending), counting sort on min(count, ORDER_BINS-1).
constexpr int ORDER_BINS = 1024;
// Sort key: (bucket >= split ? 1 : 0, size class).  With split < nb the order array comes out as
// [buckets of the first window group by falling population][buckets of the second group by falling population],
// which lets the engine accumulate and reduce the two groups in two overlapping waves (see submit_msm).
constexpr int ORDER_KEYS = 2 * ORDER_BINS;
static_assert(ORDER_BINS == (int)ITEMS_CLASSES, "the work list's population classes are the ordering keys");
// osh: populations are classed in steps of 2^osh (0 except where buckets hold more than ~1000 entries AND the accumulation runs its
// work list, which needs every population it serves inside a class: engine.hip)
MSM_D uint32_t order_key(uint32_t b, uint32_t cnt, uint32_t split, uint32_t osh) {
    return (b >= split ? (uint32_t)ORDER_BINS : 0u) + ((ORDER_BINS - 1) - min(cnt >> osh, (uint32_t)(ORDER_BINS - 1)));   // big buckets first
}
// ------------------------------------------------------------------------------------------
// fine sort of one (window, coarse bin) span: bucket count / offset tables and the index lists grouped by bucket
// ------------------------------------------------------------------------------------------
constexpr int FINE_ITEMS = 37;                         // entries per lane held in registers: nine 16-byte vectors + one head / tail entry
constexpr int FINE_CAP = 36 * SORT_BLOCK;              // 9216: spans up to this size are read once by one 256-lane workgroup
// Round 4: the same kernel with 1024 lanes reads spans up to 36864 entries once (147 KB of LDS for the permuted span: one workgroup
// per CU) -- n = 2^23 at 256 coarse bins, n = 2^24 at 512, the merged windows of a prepared table at n = 2^20: sizes that used to go
// through the chunk-parallel kernels (three passes, 16-byte runs).  `cap` (a launch parameter of the k_big_* kernels) = FINE_CAP or this.
constexpr int FINE_BLOCK_WIDE = 1024;
constexpr int FINE_CAP_WIDE = 36 * FINE_BLOCK_WIDE;

// LONG_HERE: spans beyond FINE_CAP are handled by this workgroup too (chunk by chunk: histogram sweep, then chunk-wise counting
// sort with stores by bucket runs) -- the form launched when no long span is expected, so that an unexpected one (adversarial
// scalars at small n) is still sorted correctly, just by a single workgroup.  Otherwise they are left to the k_big_* kernels.
template <class E, bool LONG_HERE, int FB = SORT_BLOCK>
__global__ void __launch_bounds__(FB) k_fine_small(const E* __restrict__ tmp, const uint32_t* __restrict__ binstart, SortGeom sg, uint32_t half,
                                                            uint32_t* __restrict__ sorted, uint32_t* __restrict__ count, uint32_t* __restrict__ offset) {
    extern __shared__ uint32_t smem_u32[];
    const uint32_t nf = 1u << sg.fbits;
    uint32_t* h = smem_u32;                 // [nf]   bucket populations, then exclusive starts
    uint32_t* out = smem_u32 + 2 * nf;      // [FINE_CAP] the span (or one chunk of it), grouped by bucket
    __shared__ uint32_t wsum[FB / 64];
    const uint32_t bin = blockIdx.x;                              // = wl * cb + coarse bin
    const uint32_t start = binstart[bin * sg.S], end = binstart[(bin + 1) * sg.S];
    const uint32_t span = end - start, t = threadIdx.x;
    // sub-bin boundaries of this span (S > 1: the entries keep the low bits of their index, the sub-bin is the rest)
    uint32_t* bnd = smem_u32 + 2 * nf + 36 * FB;                 // [S + 1]
    if (sg.S > 1) {
        for (uint32_t i = t; i <= sg.S; i += FB) bnd[i] = binstart[bin * sg.S + i];
        __syncthreads();
    }
    auto high_bits = [&](uint32_t pos) -> uint32_t {              // of the entry at position `pos` of the span
        return sg.S > 1 ? slice_of_position(bnd, sg.S, start + pos) << sg.idxbits : 0u;
    };
    const uint32_t wl = bin / sg.cb, cbin = bin % sg.cb;
    const uint32_t gb0 = wl * half + (cbin << sg.fbits);
    if (span > (uint32_t)(36 * FB)) {
        if constexpr (LONG_HERE) {
            uint32_t* th = smem_u32 + nf;   // [nf] chunk: counts, then exclusive starts inside the chunk
            for (uint32_t i = t; i < nf; i += FB) h[i] = 0;
            __syncthreads();
            for (uint32_t p = start + t; p < end; p += FB) atomicAdd(&h[entry_fine(tmp[p], sg.idxbits)], 1u);
            __syncthreads();
            for (uint32_t i = t; i < nf; i += FB) count[gb0 + i] = h[i];
            __syncthreads();
            block_exclusive_scan<FB>(h, nf, wsum);
            for (uint32_t i = t; i < nf; i += FB) offset[gb0 + i] = start + h[i];
            // h[f]: next free position of bucket f relative to `start`
            for (uint32_t cbeg = start; cbeg < end; cbeg += 36 * FB) {
                const uint32_t cn = min((uint32_t)(36 * FB), end - cbeg);
                for (uint32_t i = t; i < nf; i += FB) th[i] = 0;
                __syncthreads();
                for (uint32_t s = t; s < cn; s += FB) {
                    const E e = tmp[cbeg + s];
                    const uint32_t f = entry_fine(e, sg.idxbits);
                    const uint32_t r = atomicAdd(&th[f], 1u);
                    sorted[start + h[f] + r] = entry_sorted(e, sg.idxbits) + high_bits(cbeg - start + s);   // h is not modified during this loop
                }
                __syncthreads();
                for (uint32_t i = t; i < nf; i += FB) h[i] += th[i];
                __syncthreads();
            }
        }
        return;
    }
    for (uint32_t i = t; i < nf; i += FB) h[i] = 0;
    __syncthreads();
    E e[FINE_ITEMS];
    uint32_t rank[FINE_ITEMS];
    // Slot s of this lane's k-th entry.  4-byte entries are read and written as 16-byte vectors (a quarter of the memory
    // instructions; 4-byte accesses run at a fraction of the 16-byte rate): the span is cut at the 16-byte boundaries of its global
    // address into a head of < 4 entries, whole vectors, and a tail of < 4; lane t takes vectors t, t + 256, ... and entry k of a lane
    // is word k % 4 of its vector k / 4.  The < 8 head / tail entries ride in the last register slot of lanes 0..7.
    constexpr bool VEC = sizeof(E) == 4;
    constexpr int NV = (FINE_ITEMS - 1) / 4;    // vectors per lane (the last slot is the head / tail entry)
    uint32_t head = 0, nvec = 0, ntail = 0;
    if constexpr (VEC) {
        head = min(span, (4u - (start & 3u)) & 3u);
        nvec = (span - head) >> 2;
        ntail = span - head - 4u * nvec;
    }
    auto slot_of = [&](int k) -> uint32_t {      // position inside the span, or 0xffffffff when the slot is empty
        if constexpr (VEC) {
            if (k == FINE_ITEMS - 1) {
                if (t < head) return t;
                if (t >= 4 && t - 4 < ntail) return head + 4u * nvec + (t - 4);
                return 0xffffffffu;
            }
            if (k >= 4 * NV) return 0xffffffffu;
            const uint32_t v = (uint32_t)(k >> 2) * FB + t;
            return v < nvec ? head + 4u * v + (uint32_t)(k & 3) : 0xffffffffu;
        } else {
            const uint32_t s2 = (uint32_t)k * FB + t;
            return s2 < span ? s2 : 0xffffffffu;
        }
    };
    if constexpr (VEC) {
        static_assert(NV * 4 * FB + 8 >= 36 * FB, "the register slots cover a span of FINE_CAP entries");
        const uint4* vp = reinterpret_cast<const uint4*>(tmp + start + head);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint32_t v = (uint32_t)j * FB + t;
            if (v < nvec) {
                const uint4 q = vp[v];
                e[4 * j] = q.x;
                e[4 * j + 1] = q.y;
                e[4 * j + 2] = q.z;
                e[4 * j + 3] = q.w;
            }
        }
        const uint32_t sx = slot_of(FINE_ITEMS - 1);
        if (sx != 0xffffffffu) e[FINE_ITEMS - 1] = tmp[start + sx];
    } else {
#pragma unroll
        for (int k = 0; k < FINE_ITEMS; ++k) {
            const uint32_t s2 = slot_of(k);
            if (s2 != 0xffffffffu) e[k] = tmp[start + s2];
        }
    }
#pragma unroll
    for (int k = 0; k < FINE_ITEMS; ++k)
        if (slot_of(k) != 0xffffffffu) rank[k] = atomicAdd(&h[entry_fine(e[k], sg.idxbits)], 1u);
    __syncthreads();
    for (uint32_t i = t; i < nf; i += FB) count[gb0 + i] = h[i];
    __syncthreads();
    block_exclusive_scan<FB>(h, nf, wsum);
    for (uint32_t i = t; i < nf; i += FB) offset[gb0 + i] = start + h[i];
#pragma unroll
    for (int k = 0; k < FINE_ITEMS; ++k)
        if (slot_of(k) != 0xffffffffu) out[h[entry_fine(e[k], sg.idxbits)] + rank[k]] = entry_sorted(e[k], sg.idxbits) + high_bits(slot_of(k));
    __syncthreads();
    if constexpr (VEC) {
        uint4* op = reinterpret_cast<uint4*>(sorted + start + head);
        for (uint32_t v = t; v < nvec; v += FB) {
            const uint32_t* src = out + head + 4u * v;
            op[v] = make_uint4(src[0], src[1], src[2], src[3]);
        }
        if (t < head) sorted[start + t] = out[t];
        if (t >= 4 && t - 4 < ntail) sorted[start + head + 4u * nvec + (t - 4)] = out[head + 4u * nvec + (t - 4)];
    } else {
        for (uint32_t s2 = t; s2 < span; s2 += FB) sorted[start + s2] = out[s2];
    }
}

// ---- spans beyond FINE_CAP (large n; the few dense bins of a partly used top window; adversarial scalars): cut into chunks of
// BIG_CHUNK entries that workgroups process in parallel.
//   k_big_plan     chunks per bin and their exclusive scan
//   k_big_desc     chunk list: desc[i] = (bin, first entry of the chunk)
//   k_big_hist     per chunk: LDS histogram of the fine key -> global bucket counts (one atomic per non-empty (chunk, bucket))
//   k_big_offsets  per big bin: offsets = bin start + exclusive scan of its bucket counts; cursors start there
//   k_big_scatter  per chunk: LDS counting sort, every (chunk, bucket) run reserves its place with one atomic on the bucket's
//                  cursor and is stored by consecutive lanes
constexpr int BIG_ITEMS = 32;
constexpr int BIG_CHUNK = BIG_ITEMS * SORT_BLOCK;   // 8192 entries (16384 were tried: no faster at n = 2^26, and 164 KB of LDS at 12 fine bits)
// The chunk size is a launch parameter (`chunk`): 8192 with 1024-lane workgroups on an idle chip; 4096 with 256-lane workgroups while
// other calls are in flight -- a 1024-lane workgroup with 82 KB of LDS can only start on a CU that has drained of accumulation waves
// (twelve of them hold 107 KB of LDS and every register), a 256-lane one with 41 KB fits beside them as soon as one wave per SIMD retires.
constexpr int BIG_CHUNK_SMALL = 4096;

__global__ void __launch_bounds__(1024) k_big_plan(const uint32_t* __restrict__ binstart, uint32_t nbins, uint32_t S, uint32_t chunk, uint32_t cap, uint32_t* __restrict__ cpre) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nbins; base += 1024) {
        const uint32_t i = base + t;
        uint32_t v = 0;
        if (i < nbins) {
            const uint32_t span = binstart[(i + 1) * S] - binstart[i * S];
            v = span > cap ? (span + chunk - 1) / chunk : 0u;   // (cap: what the fine-sort kernel of this launch reads in one go)
        }
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(incl, d, 64);
            if ((int)(t & 63u) >= d) incl += o;
        }
        if ((t & 63u) == 63u) wsum[t >> 6] = incl;
        __syncthreads();
        uint32_t run = carry_s + incl - v;
        for (uint32_t q = 0; q < (t >> 6); ++q) run += wsum[q];
        if (i < nbins) cpre[i] = run;
        __syncthreads();
        if (t == 1023) carry_s = run + v;
        __syncthreads();
    }
    if (t == 0) cpre[nbins] = carry_s;
}
__global__ void __launch_bounds__(256) k_big_desc(const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ cpre, uint32_t nbins, uint32_t S,
                                                   uint32_t chunk, uint2* __restrict__ desc) {
    const uint32_t total = cpre[nbins];
    for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
        uint32_t lo = 0, hi = nbins;   // largest bin with cpre[bin] <= it (bins without chunks repeat the value: take the last)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cpre[mid] <= it) lo = mid;
            else hi = mid;
        }
        desc[it] = make_uint2(lo, binstart[lo * S] + (it - cpre[lo]) * chunk);
    }
}
template <class E>
__global__ void __launch_bounds__(SORT_BLOCK) k_big_hist(const E* __restrict__ tmp, const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ cpre,
                                                          const uint2* __restrict__ desc, SortGeom sg, uint32_t half, uint32_t chunk, uint32_t* __restrict__ count) {
    extern __shared__ uint32_t smem_u32[];
    uint32_t* h = smem_u32;   // [nf]
    const uint32_t nf = 1u << sg.fbits, total = cpre[sg.nbins], t = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < total; it += gridDim.x) {
        const uint2 d = desc[it];
        const uint32_t end = min(binstart[(d.x + 1) * sg.S], d.y + chunk);
        for (uint32_t i = t; i < nf; i += SORT_BLOCK) h[i] = 0;
        __syncthreads();
        for (uint32_t p = d.y + t; p < end; p += SORT_BLOCK) atomicAdd(&h[entry_fine(tmp[p], sg.idxbits)], 1u);
        __syncthreads();
        const uint32_t gb0 = (d.x / sg.cb) * half + ((d.x % sg.cb) << sg.fbits);
        for (uint32_t i = t; i < nf; i += SORT_BLOCK) {
            const uint32_t v = h[i];
            if (v) atomicAdd(&count[gb0 + i], v);
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SORT_BLOCK) k_big_offsets(const uint32_t* __restrict__ binstart, SortGeom sg, uint32_t half, uint32_t cap, const uint32_t* __restrict__ count,
                                                             uint32_t* __restrict__ offset, uint32_t* __restrict__ bcursor) {
    extern __shared__ uint32_t smem_u32[];
    uint32_t* h = smem_u32;   // [nf]
    __shared__ uint32_t wsum[SORT_BLOCK / 64];
    const uint32_t bin = blockIdx.x, start = binstart[bin * sg.S], t = threadIdx.x, nf = 1u << sg.fbits;
    if (binstart[(bin + 1) * sg.S] - start <= cap) return;
    const uint32_t gb0 = (bin / sg.cb) * half + ((bin % sg.cb) << sg.fbits);
    for (uint32_t i = t; i < nf; i += SORT_BLOCK) h[i] = count[gb0 + i];
    __syncthreads();
    block_exclusive_scan(h, nf, wsum);
    for (uint32_t i = t; i < nf; i += SORT_BLOCK) {
        offset[gb0 + i] = start + h[i];
        bcursor[gb0 + i] = start + h[i];
    }
}
// (1024 lanes x 8 entries per chunk: the 82 KB of staging allow one workgroup per CU, and four waves per CU left the kernel waiting
// on its own loads: 11.2 ms for 7 + 3.5 GB at n = 2^26)
constexpr int BIGS_BLOCK = 1024;
template <class E, int BIGS_BLOCK, int CHUNK>
__global__ void __launch_bounds__(BIGS_BLOCK) k_big_scatter(const E* __restrict__ tmp, const uint32_t* __restrict__ binstart, const uint32_t* __restrict__ cpre,
                                                             const uint2* __restrict__ desc, SortGeom sg, uint32_t half, uint32_t* __restrict__ bcursor,
                                                             uint32_t* __restrict__ sorted) {
    constexpr int BIGS_ITEMS = CHUNK / BIGS_BLOCK;
    extern __shared__ uint32_t smem_u32[];
    const uint32_t nf = 1u << sg.fbits;
    uint32_t* th = smem_u32;                // [nf] chunk: counts, then exclusive starts inside the staged chunk
    uint32_t* gb = smem_u32 + nf;           // [nf] global position of the chunk's run of each bucket
    uint32_t* stage = smem_u32 + 2 * nf;    // [CHUNK]
    uint32_t* spos = stage + CHUNK;     // [CHUNK]
    uint32_t* bnd = spos + CHUNK;       // [S + 1] sub-bin boundaries of the chunk's parent span (S > 1)
    __shared__ uint32_t wsum[BIGS_BLOCK / 64];
    const uint32_t total = cpre[sg.nbins], t = threadIdx.x;
    // XCD-aware order: workgroups go to the 8 XCDs round-robin, and the chunks of one span write 16-byte runs into the same ~1 MB of
    // `sorted` (2048 buckets x 4 entries per chunk at c = 20).  Spread over eight L2s those partial lines each go to HBM on their own;
    // with XCD x taking the x-th eighth of the chunk list, consecutive chunks of a span meet in ONE L2 and leave it as whole lines
    // (the grid is a multiple of 8).
    const uint32_t nx = 8, per = (total + nx - 1) / nx, x = blockIdx.x % nx, slots = gridDim.x / nx;
    for (uint32_t q = blockIdx.x / nx; q < per; q += slots) {
        const uint32_t it = x * per + q;
        if (it >= total) break;   // (uniform over the workgroup)
        const uint2 d = desc[it];
        const uint32_t cn = min(binstart[(d.x + 1) * sg.S], d.y + CHUNK) - d.y;
        const uint32_t gb0 = (d.x / sg.cb) * half + ((d.x % sg.cb) << sg.fbits);
        for (uint32_t i = t; i < nf; i += BIGS_BLOCK) th[i] = 0;
        if (sg.S > 1)
            for (uint32_t i = t; i <= sg.S; i += BIGS_BLOCK) bnd[i] = binstart[d.x * sg.S + i];
        __syncthreads();
        E e[BIGS_ITEMS];
        uint32_t rank[BIGS_ITEMS];
#pragma unroll
        for (int k = 0; k < BIGS_ITEMS; ++k) {
            const uint32_t s = k * BIGS_BLOCK + t;
            if (s < cn) e[k] = tmp[d.y + s];
        }
#pragma unroll
        for (int k = 0; k < BIGS_ITEMS; ++k) {
            const uint32_t s = k * BIGS_BLOCK + t;
            if (s < cn) rank[k] = atomicAdd(&th[entry_fine(e[k], sg.idxbits)], 1u);
        }
        __syncthreads();
        for (uint32_t i = t; i < nf; i += BIGS_BLOCK) {
            const uint32_t v = th[i];
            gb[i] = v ? atomicAdd(&bcursor[gb0 + i], v) : 0u;
        }
        __syncthreads();
        block_exclusive_scan<BIGS_BLOCK>(th, nf, wsum);
#pragma unroll
        for (int k = 0; k < BIGS_ITEMS; ++k) {
            const uint32_t s = k * BIGS_BLOCK + t;
            if (s < cn) {
                const uint32_t f = entry_fine(e[k], sg.idxbits), slot = th[f] + rank[k];
                stage[slot] = entry_sorted(e[k], sg.idxbits) + (sg.S > 1 ? slice_of_position(bnd, sg.S, d.y + s) << sg.idxbits : 0u);
                spos[slot] = gb[f] + rank[k];
            }
        }
        __syncthreads();
        for (uint32_t s = t; s < cn; s += BIGS_BLOCK) sorted[spos[s]] = stage[s];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// ordering of the buckets by population
// ------------------------------------------------------------------------------------------
// obins[k] += number of buckets with ordering key k.  (Tallying the keys inside the fine-sort kernels, which hold the populations in
// LDS, was tried: the extra barriers and the 2048-word scratch cost k_fine_small more than this 8-us pass saves.)
__global__ void __launch_bounds__(1024) k_order_hist(const uint32_t* __restrict__ count, uint32_t nb, uint32_t split, uint32_t osh, uint32_t* __restrict__ obins) {
    __shared__ uint32_t h[ORDER_KEYS];
    h[threadIdx.x] = 0;
    h[threadIdx.x + ORDER_BINS] = 0;
    __syncthreads();
    for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) atomicAdd(&h[order_key(b, count[b], split, osh)], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&obins[threadIdx.x], h[threadIdx.x]);
    if (h[threadIdx.x + ORDER_BINS]) atomicAdd(&obins[threadIdx.x + ORDER_BINS], h[threadIdx.x + ORDER_BINS]);
}
// order[]: the buckets by falling population.  obins[k] = number of buckets with key k (k_order_hist); every
// workgroup scans those 2048 counters itself (an exclusive prefix: where each key's buckets start), counts the keys of its own span of
// buckets in LDS, reserves their places with one atomic per key on ocur[k], and writes its buckets there.
__global__ void __launch_bounds__(1024) k_order_scatter(const uint32_t* __restrict__ count, uint32_t nb, uint32_t split, uint32_t osh, const uint32_t* __restrict__ obins,
                                                         uint32_t* __restrict__ ocur, uint32_t* __restrict__ order) {
    __shared__ uint32_t h[ORDER_KEYS];
    __shared__ uint32_t basep[ORDER_KEYS];
    __shared__ uint32_t wsum[16];
    const uint32_t t = threadIdx.x;
    // exclusive prefix of the key populations: two keys per lane (2t, 2t+1)
    const uint32_t a = obins[2 * t], b = obins[2 * t + 1], v = a + b;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, 64);
        if ((int)(t & 63u) >= d) incl += o;
    }
    if ((t & 63u) == 63u) wsum[t >> 6] = incl;
    h[t] = 0;
    h[t + ORDER_BINS] = 0;
    __syncthreads();
    uint32_t excl = incl - v;
    for (uint32_t q = 0; q < (t >> 6); ++q) excl += wsum[q];
    // each workgroup owns a contiguous span of buckets
    uint32_t per = (nb + gridDim.x - 1) / gridDim.x;
    uint32_t lo = blockIdx.x * per, hi = min(nb, lo + per);
    for (uint32_t bk = lo + t; bk < hi; bk += blockDim.x) atomicAdd(&h[order_key(bk, count[bk], split, osh)], 1u);
    __syncthreads();
    basep[2 * t] = excl + (h[2 * t] ? atomicAdd(&ocur[2 * t], h[2 * t]) : 0u);
    basep[2 * t + 1] = excl + a + (h[2 * t + 1] ? atomicAdd(&ocur[2 * t + 1], h[2 * t + 1]) : 0u);
    __syncthreads();
    h[t] = 0;
    h[t + ORDER_BINS] = 0;
    __syncthreads();
    for (uint32_t bk = lo + t; bk < hi; bk += blockDim.x) {
        uint32_t key = order_key(bk, count[bk], split, osh);
        uint32_t p = basep[key] + atomicAdd(&h[key], 1u);
        order[p] = bk;
    }
}

// The two prefix tables of the accumulation's work list (bucket_kernels.hpp), from the class populations the ordering pass has counted
// (first window group only: the work list is not used when a lone call is split in two groups).
// itab[0 .. 1024]: items before class k (the total at [1024]); itab[1025 .. 2049]: buckets before class k (all buckets at [2049])
__global__ void __launch_bounds__(1024) k_items_plan(const uint32_t* __restrict__ obins, uint32_t T, uint32_t osh, uint32_t* __restrict__ itab) {
    __shared__ uint32_t wsum_i[16], wsum_b[16];
    const uint32_t t = threadIdx.x, nbk = obins[t], nit = nbk * items_ni(t, T, osh);
    uint32_t inc_i = nit, inc_b = nbk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t oi = __shfl_up(inc_i, d, 64), ob = __shfl_up(inc_b, d, 64);
        if ((int)(t & 63u) >= d) {
            inc_i += oi;
            inc_b += ob;
        }
    }
    if ((t & 63u) == 63u) {
        wsum_i[t >> 6] = inc_i;
        wsum_b[t >> 6] = inc_b;
    }
    __syncthreads();
    uint32_t ex_i = inc_i - nit, ex_b = inc_b - nbk;
    for (uint32_t q = 0; q < (t >> 6); ++q) {
        ex_i += wsum_i[q];
        ex_b += wsum_b[q];
    }
    itab[t] = ex_i;
    itab[ITEMS_CLASSES + 1u + t] = ex_b;
    if (t == ITEMS_CLASSES - 1u) {
        itab[ITEMS_CLASSES] = ex_i + nit;
        itab[2u * ITEMS_CLASSES + 1u] = ex_b + nbk;
    }
}
}  // namespace msm
