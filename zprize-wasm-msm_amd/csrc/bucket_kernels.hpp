// bucket_kernels.hpp -- the ALU-bound kernels of the engine and their launchers: bucket accumulation (one lane per bucket; the work
// list: lanes per bucket in proportion to its population; slices; the tail kernels for over-long buckets) and bucket reduction.
// What they replace: the reference's table walk `T[d-1] += P_i` (wasmcurves/src/build_multiexp.js:186-233) and reduceTable (:373-461).
// Each coordinate field gets its own translation units (bucket_*.hip) so the heavy instantiations compile in parallel; engine.hip
// (engine_call.hpp) only sees the launcher declarations.  Every point addition in these kernels is expanded inline: no kernel of the
// MSM path carries a private segment beyond a few spilled registers (see tadd below and csrc/engine_ctx.hpp "scratch").
#pragma once
#include "engine_common.hpp"
#include "curve_u.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------
// stage 4: bucket accumulation -- one lane per bucket.
// A lane adds at most `seg` entries of its bucket (seg = max(32, 2x the mean population): random inputs
// practically never exceed it).  A longer bucket -- adversarial scalars, e.g. all equal -- is cut into further
// segments that the lane appends to a work list; k_accumulate_tail sums those segments with one
// lane each and k_combine_tail folds them back into the bucket.  With no long bucket the two tail
// kernels find empty lists and return.
// ------------------------------------------------------------------------------------------
struct TailCtl {
    uint32_t nwork;   // segments queued
    uint32_t nbig;    // buckets that queued segments
    uint32_t nhot;    // of which hot (engine_common.hpp): their slots in the hot list
    uint32_t pad;
};
MSM_D uint32_t* hot_list(TailCtl* ctl) { return reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ctl) + HOT_LIST_OFFSET); }
MSM_D const uint32_t* hot_list(const TailCtl* ctl) { return reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(ctl) + HOT_LIST_OFFSET); }
MSM_D uint8_t* hot_part(const TailCtl* ctl) { return const_cast<uint8_t*>(reinterpret_cast<const uint8_t*>(ctl)) + HOT_PART_OFFSET; }
// One lane queues what its bucket holds beyond `done` entries as segments of `st` entries: work items (bucket, first entry, length),
// an entry in the big-bucket list (bucket, first segment, segments, hot slot + 1 or 0).
// (reserve: the counters and the big-bucket entry, returns the first work item and their number; the items themselves are written by
// queue_segments below or, in k_accumulate, by the whole wave)
MSM_D uint32_t reserve_segments(TailCtl* __restrict__ ctl, uint4* __restrict__ biglist, uint32_t b, uint32_t cnt, uint32_t done, uint32_t st, bool may_be_hot,
                                uint32_t* extra_out) {
    const uint32_t extra = (cnt - done + st - 1u) / st;
    const uint32_t base = atomicAdd(&ctl->nwork, extra);
    const uint32_t bi = atomicAdd(&ctl->nbig, 1u);
    uint32_t hot = 0;
    if (may_be_hot && extra >= HOT_MIN_SEGS) {
        const uint32_t h = atomicAdd(&ctl->nhot, 1u);
        if (h < HOT_MAX) {
            hot = h + 1u;
            hot_list(ctl)[h] = bi;
        }
    }
    biglist[bi] = make_uint4(b, base, extra, hot);
    *extra_out = extra;
    return base;
}
MSM_D uint4 segment_item(uint32_t b, uint32_t cnt, uint32_t done, uint32_t st, uint32_t j) { return make_uint4(b, done + j * st, min(st, cnt - done - j * st), 0u); }
MSM_D void queue_segments(TailCtl* __restrict__ ctl, uint4* __restrict__ work, uint4* __restrict__ biglist, uint32_t b, uint32_t cnt, uint32_t done,
                          uint32_t st, bool may_be_hot) {
    uint32_t extra;
    const uint32_t base = reserve_segments(ctl, biglist, b, cnt, done, st, may_be_hot, &extra);
    for (uint32_t j = 0; j < extra; ++j) work[base + j] = segment_item(b, cnt, done, st, j);
}

// Buckets are held in the unsaturated representation of field_u.hpp / curve_u.hpp (BU<F>); bases arrive in
// the packed image of the reference and are re-sliced on load (no arithmetic).
template <class F>
using BU = XYZZU<typename UKind<F>::type>;

// With xbeta != nullptr (GLV path of the engine) the index lists address 2n virtual points: item 2i is base i, item 2i+1 its
// endomorphism image (xbeta[i], y_i).
template <class F>
MSM_D Affine<F> load_point(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, uint32_t idx) {
    Affine<F> p;
    if (xbeta) {
        const uint8_t* pb = bases + (size_t)(idx >> 1) * sizeof(Affine<F>);
        const uint8_t* px = (idx & 1u) ? xbeta + (size_t)(idx >> 1) * sizeof(F) : pb;
        p.x = load16<F>(px);
        p.y = load16<F>(pb + sizeof(F));
    } else {
        p = load16<Affine<F>>(bases + (size_t)idx * sizeof(Affine<F>));
    }
    return p;
}
// (Requesting the point of entry k+1 before the addition of entry k was measured: no gain at 2 waves/SIMD, and at 3 waves
// the 24 extra registers spill -- the gather latency is already covered by the other waves' ~4600 VALU instructions.  So was
// touching the next point's lines with two unused dword loads, which costs two registers: 5-10 % SLOWER at every size, n = 2^20
// to 2^26, both curves -- the accumulation at large n is not waiting for gather latency.)
// For a lane-paired kind (UG2P) both lanes of a pair walk the same entries; each loads its own component of x and y.
template <class F>
MSM_D BU<F> accumulate_span(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted, uint32_t first,
                            uint32_t len) {
    using K = typename UKind<F>::type;
    BU<F> acc = BU<F>::inf();
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t e = sorted[first + k];
        if constexpr (K::LANES == 1) {
            const Affine<F> p = load_point<F>(bases, xbeta, e & 0x7fffffffu);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&p);
            if (!words_all_zero(w, 2 * K::WORDS)) {   // else affine infinity (build_curve_jacobian_a0.js:124-171)
                auto x2 = K::load(w);
                auto y2 = fcneg(K::load(w + K::WORDS), (e >> 31) != 0);
                BU<F>::madd(acc, x2, y2);
            }
        } else {
            using B = typename F::B;                          // base field of the packed Fp2 coordinate: x = c0 || c1, y = c0 || c1
            const uint8_t* pp = bases + (size_t)(e & 0x7fffffffu) * sizeof(Affine<F>) + pair_parity() * sizeof(B);
            const B xo = load16<B>(pp);
            const B yo = load16<B>(pp + sizeof(F));
            const uint32_t z = (words_all_zero(xo.v, B::N) && words_all_zero(yo.v, B::N)) ? 1u : 0u;
            if (!(z & pair_xchg(z))) {                        // else all four components zero
                typename K::In x2{u_from_packed<typename K::Params>(xo.v)};
                typename K::In yl{u_from_packed<typename K::Params>(yo.v)};
                auto y2 = fcneg(yl, (e >> 31) != 0);
                BU<F>::madd(acc, x2, y2);
            }
        }
    }
    return acc;
}

// The same for a whole single-wave workgroup of one-lane kinds, with the entry lists staged through LDS.  Lane l walks its own list
// sorted[first_l ..): read lane by lane that is one 4-byte load per addition from 64 different lines, each line wanted again 31
// additions later -- by then evicted from the 32 KB L1 (768 such streams per CU) and, at large n, from L2 as well: the counters show
// ~2 L2 requests per addition and twice the algorithmic bytes from HBM.  Here the wave fetches 32 entries of every lane's list at a
// time, two lists per load instruction (one 128-byte line each), into a [64][33] tile; every lane then takes its entries from LDS
// (stride 33: conflict-free), and the point gather no longer waits behind a dependent load of the entry.
constexpr uint32_t STAGE_T = 32, STAGE_STRIDE = 33, STAGE_WORDS = 64 * STAGE_STRIDE + 128;   // the tile, then first[64] and len[64]
template <class F>
MSM_D BU<F> accumulate_span_staged(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted, uint32_t first,
                                   uint32_t len, uint32_t* __restrict__ tile) {
    using K = typename UKind<F>::type;
    static_assert(K::LANES == 1, "one-lane kinds only");
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* meta = tile + 64 * STAGE_STRIDE;   // the lists' starts and lengths live in LDS, not in registers, across the additions
    meta[lane] = first;
    meta[64 + lane] = len;
    uint32_t longest = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
    longest = __builtin_amdgcn_readfirstlane(longest);
    BU<F> acc = BU<F>::inf();
    // one loop with a wave-uniform trip count (a wave runs until its longest lane is done anyway): counter and refill test are scalar
    for (uint32_t k = 0; k < longest; ++k) {
        if ((k & (STAGE_T - 1u)) == 0) {
            __syncthreads();
            const uint32_t half = lane >> 5, sub = lane & 31u, j = k + sub;
#pragma unroll 8
            for (uint32_t i = 0; i < 32; ++i) {
                const uint32_t src = 2 * i + half;
                if (j < meta[64 + src]) tile[src * STAGE_STRIDE + sub] = sorted[meta[src] + j];
            }
            __syncthreads();
        }
        if (k < meta[64 + lane]) {
            const uint32_t e = tile[lane * STAGE_STRIDE + (k & (STAGE_T - 1u))];
            const Affine<F> pt = load_point<F>(bases, xbeta, e & 0x7fffffffu);
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&pt);
            if (!words_all_zero(w, 2 * K::WORDS)) {
                auto x2 = K::load(w);
                auto y2 = fcneg(K::load(w + K::WORDS), (e >> 31) != 0);
                BU<F>::madd(acc, x2, y2);
            }
        }
    }
    return acc;
}

// Full additions inside the reduction: expanded inline for the G1 fields (the kernel runs one wave per
// SIMD and is latency-bound; calls would spill the three live points around every addition), out of
// line for Fp2 where the inlined body would be ~3x larger.
template <class F>
struct InlineAdds {
    static constexpr bool value = false;
};
template <class P>
struct InlineAdds<Fp<P, true>> {
    static constexpr bool value = true;
};
// additions expanded inline in the reduction / tail-combine kernels (G1 always; experiment: G2 too)
template <class F>
struct ReduceInline {
#ifdef MSM_G2_REDUCE_INLINE
    static constexpr bool value = true;
#else
    static constexpr bool value = InlineAdds<F>::value;
#endif
};
// The tail and join kernels (k_join_slices[_tree], k_combine_tail[_lanes]) add with the addition EXPANDED INLINE as well (round 4).  Called
// out of line, the operands of an addition live in the caller's private segment (three points: 450-1030 bytes per lane), and the HSA
// runtime sizes a queue's scratch for the whole chip -- bytes per lane x 64 lanes x 8192 wave slots: 0.5 GB per hardware queue for
// BLS12-381 G2, allocated at the first dispatch and kept (tools/microbench/scratch_dispatch.hip, profiles/r04_scratch_dispatch.txt);
// when that allocation fails on a nearly full device the runtime aborts the process.  Inline, these kernels need no scratch at all.
template <class F>
MSM_D BU<F> tadd(const BU<F>& a, const BU<F>& b) {
    return BU<F>::add(a, b);
}
// Occupancy target of the accumulation (waves per SIMD): 3 for G1 -- the kernel needs 165 VGPRs, three waves keep the
// v_mad_u64_u32 pipe fed through the point loads; the Fp2 kernels need all 512 registers of a SIMD lane pair.
#ifndef MSM_ACC_WAVES
#define MSM_ACC_WAVES 3
#endif
#ifndef MSM_ACC_STAGED
#define MSM_ACC_STAGED 1   // entry lists of the G1 accumulation staged through LDS (accumulate_span_staged)
#endif
template <class F>
struct AccWaves {
    static constexpr int value = InlineAdds<F>::value ? MSM_ACC_WAVES : (UKind<F>::type::LANES == 2 ? 2 : 1);
};
// VARIANT only makes the symbols of the two builds of this kernel distinct: 1 = translation unit compiled with
// MSM_U_CHAINED_MADS (throughput: needs >= 2-3 waves per SIMD), 0 = the compiler's two interleaved chains (a launch that
// puts a single wave on a SIMD -- few owned windows, small n -- runs them almost twice as fast).
template <class F, int VARIANT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(AccWaves<F>::value))) k_accumulate(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted,
                                                     const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                                     const uint32_t* __restrict__ order, uint32_t nb, uint8_t* __restrict__ buckets, uint32_t seg, uint32_t seg_t,
                                                     TailCtl* __restrict__ ctl, uint4* __restrict__ work, uint4* __restrict__ biglist) {
    constexpr uint32_t T = UKind<F>::type::LANES;   // lanes per bucket (2: Fp2 components on a lane pair)
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t r = gt / T, sub = gt % T;
    uint32_t b = 0, cnt = 0, off = 0;
    BU<F> acc;
#if MSM_ACC_STAGED
    if constexpr (T == 1) {   // (launched as single-wave workgroups: MSM_ACC_BLOCK)
        __shared__ uint32_t tile[STAGE_WORDS];
        if (r < nb) {
            b = order[r];
            cnt = count[b];
            off = offset[b];
        }
        // (a hot bucket's own lane stops after seg_t entries: the rest is cut into segments of that length for the tail kernels)
        acc = accumulate_span_staged<F>(bases, xbeta, sorted, off, min(cnt, cnt > HOT_FACTOR * seg ? seg_t : seg), tile);
        if (r >= nb) return;
    } else
#endif
    {
        if (r >= nb) return;
        b = order[r];
        cnt = count[b];
        off = offset[b];
        acc = accumulate_span<F>(bases, xbeta, sorted, off, min(cnt, cnt > HOT_FACTOR * seg ? seg_t : seg));
    }
    store16(buckets + ((size_t)b * T + sub) * sizeof(BU<F>), acc);
    // what the bucket holds beyond its lane's share is queued for the tail kernels -- the work items of a hot bucket are tens of thousands:
    // every lane of the wave that is still here writes its share of them
    const uint32_t cap = cnt > HOT_FACTOR * seg ? seg_t : seg;
    const bool need = cnt > cap && sub == 0;
    uint32_t q_base = 0, q_extra = 0;
    if (need) q_base = reserve_segments(ctl, biglist, b, cnt, cap, cap, cap != seg, &q_extra);
    const uint64_t active = __ballot(1), needers = __ballot(need);
    if (needers == 0) return;
    const uint32_t lane = threadIdx.x & 63u, nact = (uint32_t)__popcll(active), myrank = (uint32_t)__popcll(active & ((1ull << lane) - 1ull));
    for (uint64_t m = needers; m != 0; m &= m - 1) {
        const int src = __ffsll((long long)m) - 1;
        const uint32_t sb = (uint32_t)__shfl((int)b, src), scnt = (uint32_t)__shfl((int)cnt, src), scap = (uint32_t)__shfl((int)cap, src),
                       sbase = (uint32_t)__shfl((int)q_base, src), sextra = (uint32_t)__shfl((int)q_extra, src);
        for (uint32_t j = myrank; j < sextra; j += nact) work[sbase + j] = segment_item(sb, scnt, scap, scap, j);
    }
}

// K lanes per bucket (G1 kinds).  A launch with few buckets -- the windows one rank of a sharded call owns, a small n -- would put one
// or two waves on each SIMD with chains as long as the bucket population (65536 buckets of 2048 entries at n = 2^26 over 8 ranks: one
// wave per SIMD walking 2048 additions, 20 us each).  Here bucket r is cut into K slices of ceil(cnt / K) entries; wave w adds slice
// w mod K of the 64 buckets [64 (w / K), +64) (the lanes of one wave stay as evenly loaded as the buckets are) and leaves its sums in
// `slices`; k_join_slices adds the K sums of every bucket.  The engine picks K so that the launch brings three waves to every SIMD.
// One wave per workgroup, like k_accumulate: with K waves per workgroup joined through LDS the waves were placed unevenly over the
// SIMDs (n = 2^26 / 8 ranks, K = 2: 30.5 ms against 20 ms for the same lanes as single-wave workgroups).
// `seg` bounds the entries of ONE lane; what a bucket holds beyond K * seg is queued for the tail kernels in segments of that length.
template <class F, int VARIANT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AccWaves<F>::value))) k_accumulate_slices(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                                            const uint32_t* __restrict__ order, uint32_t nb, uint32_t K, uint8_t* __restrict__ slices, uint32_t seg) {
    constexpr uint32_t T = UKind<F>::type::LANES, PER_WAVE = 64u / T;   // a lane-paired kind (G2) puts 32 buckets on a wave
    const uint32_t s = blockIdx.x % K, r = (blockIdx.x / K) * PER_WAVE + threadIdx.x / T, sub = threadIdx.x % T;
    uint32_t from = 0, len = 0;
    if (r < nb) {
        const uint32_t b = order[r], cnt = count[b];
        const uint32_t tot = min(cnt, K * seg), chunk = (tot + K - 1u) / K, first = min(s * chunk, tot);
        from = offset[b] + first;
        len = min(chunk, tot - first);
    }
    BU<F> acc;
#if MSM_ACC_STAGED
    if constexpr (T == 1) {
        __shared__ uint32_t tile[STAGE_WORDS];
        acc = accumulate_span_staged<F>(bases, xbeta, sorted, from, len, tile);
    } else
#endif
    {
        acc = accumulate_span<F>(bases, xbeta, sorted, from, len);   // (both lanes of a pair hold the same from / len)
    }
    if (r >= nb) return;
    store16(slices + (((size_t)s * nb + r) * T + sub) * sizeof(BU<F>), acc);
}
template <class F>
__global__ void __launch_bounds__(64) k_join_slices(const uint32_t* __restrict__ count, const uint32_t* __restrict__ order, uint32_t nb, uint32_t K,
                                                     const uint8_t* __restrict__ slices, uint8_t* __restrict__ buckets, uint32_t seg,
                                                     TailCtl* __restrict__ ctl, uint4* __restrict__ work, uint4* __restrict__ biglist) {
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t gt = blockIdx.x * 64u + threadIdx.x, r = gt / T, sub = gt % T;
    if (r >= nb) return;   // (both lanes of a pair leave together)
    BU<F> acc = load16<BU<F>>(slices + ((size_t)r * T + sub) * sizeof(BU<F>));
    for (uint32_t s = 1; s < K; ++s) acc = tadd<F>(acc, load16<BU<F>>(slices + (((size_t)s * nb + r) * T + sub) * sizeof(BU<F>)));
    const uint32_t b = order[r], cnt = count[b];
    store16(buckets + ((size_t)b * T + sub) * sizeof(BU<F>), acc);
    if (cnt > K * seg && sub == 0) queue_segments(ctl, work, biglist, b, cnt, K * seg, seg, false);
}

// The join for many slices per bucket (K a power of two, 8..64): a 256-lane workgroup takes 256 / K buckets, lane (s, r) loads slice s
// of bucket r, and the slices are folded in log2 K rounds through LDS -- 6 dependent additions instead of 63 at K = 64 (a small
// n in 8-bit windows: 4224 buckets of hundreds of entries, where the serial join was most of the accumulation stage).
template <class F>
__global__ void __launch_bounds__(256) k_join_slices_tree(const uint32_t* __restrict__ count, const uint32_t* __restrict__ order, uint32_t nb, uint32_t K,
                                                          const uint8_t* __restrict__ slices, uint8_t* __restrict__ buckets, uint32_t seg,
                                                          TailCtl* __restrict__ ctl, uint4* __restrict__ work, uint4* __restrict__ biglist) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    BU<F>* sh = reinterpret_cast<BU<F>*>(smem);   // [K][B][T]
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t B = 256u / (K * T), sub = threadIdx.x % T, rb = (threadIdx.x / T) % B, s = threadIdx.x / (T * B), r = blockIdx.x * B + rb;
    const bool live = r < nb;
    BU<F> acc = live ? load16<BU<F>>(slices + (((size_t)s * nb + r) * T + sub) * sizeof(BU<F>)) : BU<F>::inf();
    for (uint32_t h = K >> 1; h >= 1; h >>= 1) {
        if (s >= h && s < 2 * h) sh[(s * B + rb) * T + sub] = acc;
        __syncthreads();
        if (s < h) acc = tadd<F>(acc, sh[((s + h) * B + rb) * T + sub]);   // (s is the same for both lanes of a pair)
        __syncthreads();
    }
    if (s != 0 || !live) return;
    const uint32_t b = order[r], cnt = count[b];
    store16(buckets + ((size_t)b * T + sub) * sizeof(BU<F>), acc);
    if (cnt > K * seg && sub == 0) queue_segments(ctl, work, biglist, b, cnt, K * seg, seg, false);
}

// ------------------------------------------------------------------------------------------
// The work list ("items", round 4): lanes per bucket in proportion to its population.
// A launch with few buckets, or with short ones, is a single round of waves whose duration is its LONGEST lane: with one lane per bucket
// that is the largest population (n = 2^16 in 14-bit windows: mean 8, largest ~22), with K slices per bucket it is largest / K, and K is
// the same for a bucket of 3 entries and one of 30.  Here every lane adds at most T entries: a bucket of cnt entries is served by
// ni = ceil(cnt / T) lanes (at most ITEMS_NI_MAX; what a skewed bucket holds beyond that goes to the tail kernels as before), each taking
// ceil(cnt / ni) of them.  No list is materialised: the buckets are already ordered by falling population (`order`, counting sort on
// the key of sort_kernels.hpp), so all buckets of one population class are consecutive ranks and get the same ni -- the item index of
// a lane, the rank of its bucket and its slice follow from two 1025-entry prefix tables over the classes (items before class k,
// buckets before class k: k_items_plan, one workgroup), found by binary search.  Lanes of buckets that need only one lane write the
// bucket itself; the others leave their sums in `items_out` and k_join_items adds the ni sums of every multi-lane bucket (the
// multi-lane buckets are a prefix of the ranks), with the addition expanded inline: no private segment.
// ------------------------------------------------------------------------------------------
// largest k in [0, 1023] with tab[k] <= g, given tab[1024] > g (tab: non-decreasing exclusive prefix, 1025 entries)
MSM_D uint32_t items_class_of(const uint32_t* __restrict__ tab, uint32_t g) {
    uint32_t lo = 0, hi = ITEMS_CLASSES;
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tab[mid] <= g) lo = mid;
        else hi = mid;
    }
    return lo;
}
// (the tables: k_items_plan in sort_kernels.hpp -- itab[0 .. 1024]: items before class k, the total at [1024]; itab[1025 .. 2049]: buckets before class k)
template <class F, int VARIANT>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AccWaves<F>::value))) k_accumulate_items(
    const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ offset,
    const uint32_t* __restrict__ count, const uint32_t* __restrict__ order, const uint32_t* __restrict__ itab, uint32_t T, uint32_t osh,
    uint8_t* __restrict__ buckets, uint8_t* __restrict__ items_out) {
    constexpr uint32_t TL = UKind<F>::type::LANES, PER_WAVE = 64u / TL;
    const uint32_t total = itab[ITEMS_CLASSES];
    if (blockIdx.x * PER_WAVE >= total) return;   // (the grid is sized for the most items the entries could make: the whole wave leaves)
    const uint32_t g = blockIdx.x * PER_WAVE + threadIdx.x / TL, sub = threadIdx.x % TL;
    const bool live = g < total;
    uint32_t from = 0, len = 0, b = 0, ni = 1;
    if (live) {
        const uint32_t k = items_class_of(itab, g), j = g - itab[k];
        ni = items_ni(k, T, osh);
        const uint32_t r = itab[ITEMS_CLASSES + 1u + k] + j / ni, s = j % ni;
        b = order[r];
        const uint32_t cnt = count[b], covered = min(cnt, ni * T), chunk = (covered + ni - 1u) / ni, first = min(s * chunk, covered);
        from = offset[b] + first;
        len = min(chunk, covered - first);
    }
    BU<F> acc;
#if MSM_ACC_STAGED
    if constexpr (TL == 1) {
        __shared__ uint32_t tile[STAGE_WORDS];
        acc = accumulate_span_staged<F>(bases, xbeta, sorted, from, len, tile);
    } else
#endif
    {
        acc = accumulate_span<F>(bases, xbeta, sorted, from, len);   // (both lanes of a pair hold the same from / len)
    }
    if (!live) return;
    if (ni == 1) store16(buckets + ((size_t)b * TL + sub) * sizeof(BU<F>), acc);
    else store16(items_out + ((size_t)g * TL + sub) * sizeof(BU<F>), acc);
}
// one (logical) lane per multi-lane bucket -- ranks [0, first rank of a one-lane class): the ni sums of the bucket, added with the
// addition expanded inline; what the bucket holds beyond ni * T entries (skewed inputs only) is queued for the tail kernels
template <class F>
__global__ void __launch_bounds__(64) k_join_items(const uint32_t* __restrict__ count, const uint32_t* __restrict__ order, const uint32_t* __restrict__ itab, uint32_t T,
                                                    uint32_t osh, const uint8_t* __restrict__ items_out, uint8_t* __restrict__ buckets, uint32_t seg, uint32_t seg_t,
                                                    TailCtl* __restrict__ ctl, uint4* __restrict__ work, uint4* __restrict__ biglist) {
    constexpr uint32_t TL = UKind<F>::type::LANES;
    const uint32_t gt = blockIdx.x * 64u + threadIdx.x, r = gt / TL, sub = gt % TL;
    const uint32_t* rtab = itab + ITEMS_CLASSES + 1u;
    if (r >= rtab[ITEMS_CLASSES]) return;   // (both lanes of a pair leave together)
    const uint32_t k = items_class_of(rtab, r), ni = items_ni(k, T, osh);
    if (ni == 1) return;
    const uint32_t g0 = itab[k] + (r - rtab[k]) * ni;
    BU<F> acc = load16<BU<F>>(items_out + ((size_t)g0 * TL + sub) * sizeof(BU<F>));
    for (uint32_t s = 1; s < ni; ++s) acc = BU<F>::add(acc, load16<BU<F>>(items_out + ((size_t)(g0 + s) * TL + sub) * sizeof(BU<F>)));
    const uint32_t b = order[r], cnt = count[b], covered = min(cnt, ni * T);
    store16(buckets + ((size_t)b * TL + sub) * sizeof(BU<F>), acc);
    if (cnt > covered && sub == 0) {
        const bool hot = cnt > HOT_FACTOR * seg;
        queue_segments(ctl, work, biglist, b, cnt, covered, hot ? seg_t : seg, hot);
    }
}

template <class F>
__global__ void __launch_bounds__(256) k_accumulate_tail(const uint8_t* __restrict__ bases, const uint8_t* __restrict__ xbeta, const uint32_t* __restrict__ sorted,
                                                          const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                                          const TailCtl* __restrict__ ctl, const uint4* __restrict__ work,
                                                          uint8_t* __restrict__ seg_out) {
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t nwork = ctl->nwork;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t item = gt / T; item < nwork; item += gridDim.x * blockDim.x / T) {
        const uint4 w = work[item];   // bucket, first entry, entries
        BU<F> acc = accumulate_span<F>(bases, xbeta, sorted, offset[w.x] + w.y, w.z);
        store16(seg_out + ((size_t)item * T + gt % T) * sizeof(BU<F>), acc);
    }
}

// A bucket's queued segments are summed by a whole workgroup (LDS tree) when they are many -- adversarial inputs: one bucket, tens of
// thousands of segments -- and by one lane when they are few: the 2^14 twice-as-dense buckets of a narrow top window, cut into a handful
// of segments each by a sliced launch, would otherwise take one workgroup iteration apiece (16384 iterations over 64 workgroups).
constexpr uint32_t TAIL_LANE_SEGS = 24;   // up to this many segments: k_combine_tail_lanes
template <class F>
__global__ void __launch_bounds__(256) k_combine_tail(const TailCtl* __restrict__ ctl, const uint4* __restrict__ biglist,
                                                       const uint8_t* __restrict__ seg_out, uint8_t* __restrict__ buckets) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    BU<F>* sh = reinterpret_cast<BU<F>*>(smem);
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t lt = threadIdx.x / T, sub = threadIdx.x % T, nlt = blockDim.x / T;   // logical lane, component, logical lanes
    const uint32_t nbig = ctl->nbig, nhot = min(ctl->nhot, HOT_MAX);
    uint8_t* part = hot_part(ctl);
    // Two kinds of work unit share ONE body (three inlined additions instead of six):
    //   phase 0: an ordinary long bucket -- its segments summed by the whole workgroup and added to the bucket;
    //   phase 1: hot buckets, first fold -- unit (slot, g) sums the segments g * nlt + lt, + HOT_GROUPS * nlt, ... of hot bucket `slot` and
    //            leaves ONE partial sum; k_combine_tail_lanes (the next launch) folds the HOT_GROUPS partials of every hot bucket.
    for (uint32_t phase = 0; phase < 2; ++phase) {
        const uint32_t nunits = phase == 0 ? nbig : nhot * HOT_GROUPS;
        for (uint32_t unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
            uint4 d;   // bucket, first segment, number of segments, hot slot + 1
            uint32_t j0, jstep;
            uint8_t* dst;
            if (phase == 0) {
                d = biglist[unit];
                if (d.z <= TAIL_LANE_SEGS || d.w != 0) continue;   // (uniform over the workgroup; few segments: k_combine_tail_lanes; hot: phase 1)
                j0 = lt;
                jstep = nlt;
                dst = buckets + ((size_t)d.x * T + sub) * sizeof(BU<F>);
            } else {
                const uint32_t slot = unit / HOT_GROUPS, g = unit % HOT_GROUPS;
                d = biglist[hot_list(ctl)[slot]];
                j0 = g * nlt + lt;
                jstep = HOT_GROUPS * nlt;
                dst = part + ((size_t)unit * T + sub) * sizeof(BU<F>);
            }
            BU<F> loc = BU<F>::inf();
            for (uint32_t j = j0; j < d.z; j += jstep) loc = tadd<F>(loc, load16<BU<F>>(seg_out + ((size_t)(d.y + j) * T + sub) * sizeof(BU<F>)));
            sh[threadIdx.x] = loc;
            __syncthreads();
            for (uint32_t s = nlt / 2; s > 0; s >>= 1) {
                if (lt < s) {
                    BU<F> a = sh[threadIdx.x];
                    BU<F> bb = sh[threadIdx.x + s * T];
                    sh[threadIdx.x] = tadd<F>(a, bb);
                }
                __syncthreads();
            }
            if (lt == 0) {
                BU<F> r = sh[threadIdx.x];
                if (phase == 0) r = tadd<F>(load16<BU<F>>(dst), r);
                store16(dst, r);
            }
            __syncthreads();
        }
    }
}
template <class F>
__global__ void __launch_bounds__(64) k_combine_tail_lanes(const TailCtl* __restrict__ ctl, const uint4* __restrict__ biglist,
                                                            const uint8_t* __restrict__ seg_out, uint8_t* __restrict__ buckets) {
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t nbig = ctl->nbig, gt = blockIdx.x * blockDim.x + threadIdx.x, sub = gt % T;
    for (uint32_t bi = gt / T; bi < nbig; bi += gridDim.x * blockDim.x / T) {   // (both lanes of a pair take the same trips)
        uint4 d = biglist[bi];
        if (d.z > TAIL_LANE_SEGS) continue;
        BU<F> acc = load16<BU<F>>(buckets + ((size_t)d.x * T + sub) * sizeof(BU<F>));
        for (uint32_t j = 0; j < d.z; ++j) acc = tadd<F>(acc, load16<BU<F>>(seg_out + ((size_t)(d.y + j) * T + sub) * sizeof(BU<F>)));
        store16(buckets + ((size_t)d.x * T + sub) * sizeof(BU<F>), acc);
    }
    // hot buckets, second fold: the HOT_GROUPS partial sums that k_combine_tail (the previous launch) left per hot bucket, by one wave;
    // the bucket's own sum joins the tree as one more leaf (lane HOT_GROUPS), so that the fold is ONE inlined addition
    __shared__ __attribute__((aligned(16))) BU<F> shh[64];
    constexpr uint32_t NLT = 64u / T;   // logical lanes of this 64-thread workgroup
    static_assert(HOT_GROUPS < NLT || T == 2, "one logical lane per partial sum and one for the bucket");
    const uint32_t nhot = min(ctl->nhot, HOT_MAX), lt = threadIdx.x / T;
    const uint8_t* part = hot_part(ctl);
    for (uint32_t slot = blockIdx.x; slot < nhot; slot += gridDim.x) {
        const uint4 d = biglist[hot_list(ctl)[slot]];
        uint8_t* dst = buckets + ((size_t)d.x * T + sub) * sizeof(BU<F>);
        if constexpr (HOT_GROUPS < NLT) {
            shh[threadIdx.x] = lt < HOT_GROUPS ? load16<BU<F>>(part + (((size_t)slot * HOT_GROUPS + lt) * T + sub) * sizeof(BU<F>))
                                               : (lt == HOT_GROUPS ? load16<BU<F>>(dst) : BU<F>::inf());
        } else {   // (a point on a lane pair: 32 logical lanes hold the 32 partials; the bucket's own sum is added to lane 0's leaf below)
            shh[threadIdx.x] = load16<BU<F>>(part + (((size_t)slot * HOT_GROUPS + lt) * T + sub) * sizeof(BU<F>));
        }
        __syncthreads();
        // (with the pair kind the tree starts one level higher: a first round adds the bucket into leaf 0 -- same code, an extra trip)
        for (uint32_t s = (HOT_GROUPS < NLT ? NLT / 2 : NLT); s > 0; s >>= 1) {
            const bool extra = s == NLT;   // pair kind only: leaf 0 += the bucket's own sum
            if (lt < s && (!extra || lt == 0)) {
                BU<F> a = shh[threadIdx.x];
                BU<F> bb = extra ? load16<BU<F>>(dst) : shh[threadIdx.x + s * T];
                shh[threadIdx.x] = tadd<F>(a, bb);
            }
            __syncthreads();
        }
        if (lt == 0) store16(dst, shh[threadIdx.x]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// stage 5: bucket reduction  S_w = sum_{k=1..half} k * B[w][k-1]
// Lane t of window w owns buckets [t*L, t*L+L): running sums give sum_j (j+1) B[t*L+j] and
// A = sum_j B[t*L+j]; adding (t*L)*A (double-and-add) makes the lane's share exact; a workgroup
// tree in LDS then leaves one partial per workgroup.
// ------------------------------------------------------------------------------------------
template <class F>
MSM_D BU<F> radd(const BU<F>& a, const BU<F>& b) {
    if constexpr (ReduceInline<F>::value) return BU<F>::add(a, b);
    else return BU<F>::add_ni(a, b);
}
template <class F>
MSM_D BU<F> rdbl(const BU<F>& a) {
    if constexpr (ReduceInline<F>::value) return BU<F>::dbl(a);
    else return BU<F>::dbl_ni(a);
}

template <class F>
__global__ void __launch_bounds__(256) k_bucket_reduce(const uint8_t* __restrict__ buckets, uint32_t half, uint32_t L, uint32_t lanes_per_window,
                                                        uint32_t wgs_per_window, uint8_t* __restrict__ partials, const uint32_t* __restrict__ offset,
                                                        const uint32_t* __restrict__ count, uint32_t gb0, uint32_t NB) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    BU<F>* sh = reinterpret_cast<BU<F>*>(smem);
    constexpr uint32_t T = UKind<F>::type::LANES;
    const uint32_t lt = threadIdx.x / T, sub = threadIdx.x % T, nlt = blockDim.x / T;
    uint32_t w = blockIdx.x / wgs_per_window;
    uint32_t wg = blockIdx.x % wgs_per_window;
    {
        // a window nothing was scattered into (the carry window above 255-bit scalars): all its buckets are infinity, and so is
        // its sum -- the whole workgroup leaves (offsets run over the windows in order; gb0 = first bucket of this launch)
        const uint32_t gb = gb0 + w * half;
        const uint32_t beg = offset[gb];
        const uint32_t end = gb + half < NB ? offset[gb + half] : offset[NB - 1] + count[NB - 1];
        if (beg == end) {
            if (lt == 0) {
                BU<F> z = BU<F>::inf();
                z.to_packed(reinterpret_cast<uint32_t*>(partials + (size_t)blockIdx.x * sizeof(XYZZ<F>)));
            }
            return;
        }
    }
    uint32_t t = wg * nlt + lt;
    BU<F> res = BU<F>::inf();
    if (t < lanes_per_window) {
        uint32_t base = t * L;
        BU<F> run = BU<F>::inf();
        BU<F> sum = BU<F>::inf();
        for (int j = (int)L - 1; j >= 0; --j) {
            if (base + j < half) {
                BU<F> bk = load16<BU<F>>(buckets + (((size_t)w * half + base + j) * T + sub) * sizeof(BU<F>));
                run = radd<F>(run, bk);
            }
            sum = radd<F>(sum, run);
        }
        // res = sum + base * run
        BU<F> m = BU<F>::inf();
        if (base != 0) {
            int top = 31 - __clz(base);
            for (int bit = top; bit >= 0; --bit) {
                m = rdbl<F>(m);
                if ((base >> bit) & 1u) m = radd<F>(m, run);
            }
        }
        res = radd<F>(sum, m);
    }
    sh[threadIdx.x] = res;
    __syncthreads();
    for (uint32_t d = nlt / 2; d > 0; d >>= 1) {
        if (lt < d) {
            BU<F> a = sh[threadIdx.x];
            BU<F> bb = sh[threadIdx.x + d * T];
            sh[threadIdx.x] = radd<F>(a, bb);
        }
        __syncthreads();
    }
    // leaves the GPU: back to the canonical packed image XYZZ<F> that the host combine reads (a lane pair writes one component each)
    if (lt == 0) sh[threadIdx.x].to_packed(reinterpret_cast<uint32_t*>(partials + (size_t)blockIdx.x * sizeof(XYZZ<F>)));
}

// ------------------------------------------------------------------------------------------
// stage 5 without a scalar multiplication ("bit planes", round 4).  With lane t of a window owning the L buckets from t*L on,
//     S_w = sum_t T_t + L * sum_t t * A_t          (T_t = sum_j (j+1) B[tL+j],  A_t = sum_j B[tL+j]: the running sums above)
//         = sum_t T_t + sum_b 2^(b + log2 L) * P_b   with   P_b = sum of A_t over the lanes t whose bit b is set,
// and the weights 2^(b + log2 L) cost nothing: the host's Horner pass over the windows doubles once per bit anyway and adds P_b when
// it passes bit (window start + b + log2 L).  What is left on the GPU are PLAIN sums: k_bucket_reduce's per-lane double-and-add by
// t*L (13 doublings + ~7 additions at L = 8: 45 % of its instructions and of its dependent chain) is gone.
//
// The unit of work is ONE WAVE -- no LDS, no barrier, nobody waits for anybody: a SIMD's multiplier is saturated by a single wave of
// these additions, so what counts is that every resident wave is busy (in a 256-lane workgroup tree three of four waves idle at the
// barrier and hold their registers; measured: 1.1 ms for the 2048 workgroups of n = 2^20 against 0.72 ms for k_bucket_reduce).
// After the running sums the 2^k lanes of the wave (k = 6; 5 where a point takes a lane pair) hold A_t and T_t in registers, partners
// are fetched with ds_bpermute, and ONE loop of k + 1 additions finishes both sums:
//  * the planes come out of an in-place butterfly over X[lane] = A_t: step s (d = 2^(k-1-s)) folds X[e] += X[e+d] for e < d -- X[0]
//    ends as the total -- while the upper halves X[d .. 2d) it leaves behind are exactly the summands of plane k-1-s and are summed in
//    place by the following steps (X[d+e] += X[d+e+m], m = d/2, d/4, ...): P_b ends in lane 2^b;
//  * the sum of T_t folds UPWARD (lane i += lane i - 2^(k-1-i')), one step ahead of the butterfly, so that its lanes (the top 2^(k-1-i'))
//    never coincide with the butterfly's (below 2^(k-1) + d/2): it ends in the last lane.
// A window of several waves is finished by k_planes_combine: one wave per (window, value).  2L + k + 1 additions per wave of 64 L
// buckets -- 2.9 per bucket at L = 8 (k_bucket_reduce: 4.9), chain 23 + 7 instead of 45.
// ------------------------------------------------------------------------------------------
template <class P>
MSM_D P lane_fetch(const P& p, uint32_t src_thread) {
    constexpr int NW = sizeof(P) / 4;
    static_assert(sizeof(P) % 4 == 0, "whole words");
    struct Words {
        uint32_t w[NW];
    };
    const Words s = __builtin_bit_cast(Words, p);
    Words r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_thread * 4u), (int)s.w[i]);
    return __builtin_bit_cast(P, r);
}
template <class P>
MSM_D P lane_select(bool first, const P& a, const P& b) {
    constexpr int NW = sizeof(P) / 4;
    struct Words {
        uint32_t w[NW];
    };
    const Words x = __builtin_bit_cast(Words, a), y = __builtin_bit_cast(Words, b);
    Words r;
#pragma unroll
    for (int i = 0; i < NW; ++i) r.w[i] = first ? x.w[i] : y.w[i];
    return __builtin_bit_cast(P, r);
}
// butterfly step over 2^klog logical lanes, in place (the lane that holds X[dst] adds): the partner of lane lt at step s, or none
MSM_D bool planes_partner(uint32_t lt, uint32_t klog, uint32_t s, bool trees, uint32_t* src) {
    const uint32_t d = (1u << klog) >> (s + 1), m = d >> 1;
    if (lt < d) {
        *src = lt + d;
        return true;
    }
    const uint32_t base = 1u << (31 - __clz(lt | 1u));   // the region [2^j, 2^(j+1)) lane lt lies in: the summands of plane j (j >= log2 d here)
    *src = lt + m;
    return trees && lt >= d && lt < (1u << klog) && lt - base < m;
}
template <class F>
MSM_D bool planes_window_empty(const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count, uint32_t gb, uint32_t half, uint32_t NB) {
    const uint32_t beg = offset[gb];
    const uint32_t end = gb + half < NB ? offset[gb + half] : offset[NB - 1] + count[NB - 1];
    return beg == end;
}
// values a wave leaves for its window when the window has several: [0] sum of T_t, [1] total, [2+b] P_b of the local lane bit b
template <class F>
__global__ void __launch_bounds__(64) k_bucket_planes(const uint8_t* __restrict__ buckets, uint32_t half, uint32_t L, uint32_t wpw, uint8_t* __restrict__ wvout,
                                                       uint8_t* __restrict__ partials, const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                                       uint32_t gb0, uint32_t NB) {
    constexpr uint32_t T = UKind<F>::type::LANES, LPW = 64u / T, KLOG = T == 1 ? 6u : 5u, NV = KLOG + 2u;
    static_assert(T == 1 || T == 2, "a point on one lane or on a lane pair");
    const uint32_t lt = threadIdx.x / T, sub = threadIdx.x % T;
    const uint32_t w = blockIdx.x / wpw;
    constexpr uint32_t np1 = 1 + KLOG;   // partials of a window that one wave covers
    if (planes_window_empty<F>(offset, count, gb0 + w * half, half, NB)) {
        // nothing was scattered into this window (the carry window above 255-bit scalars): every sum is infinity
        if (wpw == 1 && lt < np1) BU<F>::inf().to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np1 + lt) * sizeof(XYZZ<F>)));
        return;   // (several waves: k_planes_combine sees the same and writes the infinities)
    }
    const uint32_t base = ((blockIdx.x % wpw) * LPW + lt) * L;
    BU<F> run = BU<F>::inf(), sum = BU<F>::inf();
    for (int j = (int)L - 1; j >= 0; --j) {
        const BU<F> bk = load16<BU<F>>(buckets + (((size_t)w * half + base + j) * T + sub) * sizeof(BU<F>));
        run = radd<F>(run, bk);
        sum = radd<F>(sum, run);
    }
    for (uint32_t i = 0; i <= KLOG; ++i) {
        // the upward fold of the T_t (i < KLOG) and butterfly step i - 1 (i >= 1): disjoint lanes
        const uint32_t h = LPW >> (i + 1);
        const bool t_on = i < KLOG && lt >= LPW - h;
        uint32_t a_src = lt;
        const bool a_on = i >= 1 && planes_partner(lt, KLOG, i - 1, true, &a_src);
        const BU<F> ft = lane_fetch(sum, ((lt - h) & (LPW - 1u)) * T + sub);
        const BU<F> fa = lane_fetch(run, (a_src & (LPW - 1u)) * T + sub);
        const BU<F> r = radd<F>(lane_select(t_on, sum, run), lane_select(t_on, ft, fa));
        if (t_on) sum = r;
        if (a_on) run = r;
    }
    // lane LPW-1: sum of T_t; lane 0: total; lane 2^b: P_b
    if (wpw == 1) {
        if (lt == LPW - 1) sum.to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np1) * sizeof(XYZZ<F>)));
        if (lt != 0 && (lt & (lt - 1)) == 0 && lt != LPW - 1) run.to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np1 + 1 + (31 - __clz(lt))) * sizeof(XYZZ<F>)));
    } else {
        uint8_t* o = wvout + (size_t)blockIdx.x * NV * T * sizeof(BU<F>);
        if (lt == LPW - 1) store16(o + sub * sizeof(BU<F>), sum);
        if ((lt & (lt - 1)) == 0) store16(o + ((size_t)(lt == 0 ? 1u : 2u + (31 - __clz(lt))) * T + sub) * sizeof(BU<F>), run);
    }
}
// The window's sums over its waves: one wave per (window, value) -- plain trees for the sum of T_t and the local planes, the butterfly
// again over the waves' totals (planes of the bits above the local ones).  A window of more waves than a wave has lanes (up to four
// times as many: qlog <= 2) gives every lane 2^qlog of them to add up first, wave q*LPW + lane for q = 0, 1, ...; the planes of the
// bits of q are plain sums again -- of the totals of the waves whose q has the bit -- and get a wave of their own (v >= NV).
// Partials of a window, in the order the host adds them: [0] weight 1, [1+b] weight 2^(b + log2 L), b < KLOG + mlog.
template <class F>
__global__ void __launch_bounds__(64) k_planes_combine(const uint8_t* __restrict__ wvout, uint32_t half, uint32_t mlog, uint8_t* __restrict__ partials,
                                                        const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count, uint32_t gb0, uint32_t NB) {
    constexpr uint32_t T = UKind<F>::type::LANES, LPW = 64u / T, KLOG = T == 1 ? 6u : 5u, NV = KLOG + 2u;
    const uint32_t lt = threadIdx.x / T, sub = threadIdx.x % T, wpw = 1u << mlog;
    const uint32_t tlog = min(mlog, KLOG), qlog = mlog - tlog;   // lanes of the tree, waves per lane
    const uint32_t np = 1 + KLOG + mlog, nvq = NV + qlog;
    const uint32_t w = blockIdx.x / nvq, v = blockIdx.x % nvq;
    if (planes_window_empty<F>(offset, count, gb0 + w * half, half, NB)) {
        if (v == 0 && lt < np) BU<F>::inf().to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np + lt) * sizeof(XYZZ<F>)));
        return;
    }
    const uint32_t vv = v < NV ? v : 1u;   // the waves of the upper planes add totals
    auto elem = [&](uint32_t q) { return load16<BU<F>>(wvout + ((((size_t)w * wpw + q * LPW + lt) * NV + vv) * T + sub) * sizeof(BU<F>)); };
    BU<F> x = BU<F>::inf();
    if (lt < (1u << tlog)) {
        if (v < NV) {
            x = elem(0);
            for (uint32_t q = 1; q < (1u << qlog); ++q) x = radd<F>(x, elem(q));
        } else {
            for (uint32_t q = 1; q < (1u << qlog); ++q)
                if ((q >> (v - NV)) & 1u) x = radd<F>(x, elem(q));
        }
    }
    for (uint32_t s = 0; s < tlog; ++s) {
        uint32_t src = lt;
        const bool on = planes_partner(lt, tlog, s, v == 1, &src);
        const BU<F> f = lane_fetch(x, (src & (LPW - 1u)) * T + sub);
        const BU<F> r = radd<F>(x, f);
        if (on) x = r;
    }
    if (v != 1) {
        if (lt == 0) x.to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np + (v == 0 ? 0u : (v < NV ? v - 1u : 1u + KLOG + tlog + (v - NV)))) * sizeof(XYZZ<F>)));
    } else if (lt != 0 && lt < (1u << tlog) && (lt & (lt - 1)) == 0) {
        x.to_packed(reinterpret_cast<uint32_t*>(partials + ((size_t)w * np + 1 + KLOG + (31 - __clz(lt))) * sizeof(XYZZ<F>)));
    }
}
template <class F>
void launch_bucket_planes(hipStream_t st, uint32_t windows, uint32_t half, uint32_t L, uint32_t mlog, const uint8_t* buckets, uint8_t* wvout, uint8_t* partials,
                          const uint32_t* offset, const uint32_t* count, uint32_t gb0, uint32_t NB) {
    constexpr uint32_t KLOG = UKind<F>::type::LANES == 1 ? 6u : 5u;
    const uint32_t wpw = 1u << mlog;
    k_bucket_planes<F><<<windows * wpw, 64, 0, st>>>(buckets, half, L, wpw, wvout, partials, offset, count, gb0, NB);
    if (wpw > 1) k_planes_combine<F><<<windows * (KLOG + 2u + (mlog > KLOG ? mlog - KLOG : 0u)), 64, 0, st>>>(wvout, half, mlog, partials, offset, count, gb0, NB);
}
// logical lanes of a wave in the plane kernels, as a power of two
template <class F>
uint32_t planes_lane_bits() {
    return UKind<F>::type::LANES == 1 ? 6u : 5u;
}

// The same reduction for G1 with every point split over a lane pair (XYZZH, curve_u.hpp): the dependent chain of a lane is 8 instead
// of 14 field products per addition and its live state halves (two such waves fit a SIMD beside an accumulation wave).  Logical lane
// lt = threadIdx / 2 owns L consecutive buckets exactly as above; `rblock` counts logical lanes.
template <class F>
__global__ void __launch_bounds__(512) k_bucket_reduce_pair(const uint8_t* __restrict__ buckets, uint32_t half, uint32_t L, uint32_t lanes_per_window,
                                                             uint32_t wgs_per_window, uint8_t* __restrict__ partials, const uint32_t* __restrict__ offset,
                                                             const uint32_t* __restrict__ count, uint32_t gb0, uint32_t NB) {
    using H = XYZZH<typename F::Params>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    H* sh = reinterpret_cast<H*>(smem);
    const uint32_t lt = threadIdx.x >> 1, nlt = blockDim.x >> 1;
    const uint32_t w = blockIdx.x / wgs_per_window, wg = blockIdx.x % wgs_per_window;
    {
        const uint32_t gb = gb0 + w * half;
        const uint32_t beg = offset[gb];
        const uint32_t end = gb + half < NB ? offset[gb + half] : offset[NB - 1] + count[NB - 1];
        if (beg == end) {   // nothing was scattered into this window: its sum is infinity
            if (lt == 0) H::inf().to_packed(reinterpret_cast<uint32_t*>(partials + (size_t)blockIdx.x * sizeof(XYZZ<F>)));
            return;
        }
    }
    const uint32_t t = wg * nlt + lt;
    H res = H::inf();
    if (t < lanes_per_window) {   // both lanes of a pair share t
        const uint32_t base = t * L;
        H run = H::inf(), sum = H::inf();
        for (int j = (int)L - 1; j >= 0; --j) {
            if (base + j < half) run = H::add(run, H::load_bucket(buckets + ((size_t)w * half + base + j) * sizeof(BU<F>)));
            sum = H::add(sum, run);
        }
        H m = H::inf();   // res = sum + base * run
        if (base != 0) {
            const int top = 31 - __clz(base);
            for (int bit = top; bit >= 0; --bit) {
                m = H::dbl(m);
                if ((base >> bit) & 1u) m = H::add(m, run);
            }
        }
        res = H::add(sum, m);
    }
    sh[threadIdx.x] = res;
    __syncthreads();
    for (uint32_t d = nlt / 2; d > 0; d >>= 1) {
        if (lt < d) {
            const H a = sh[threadIdx.x], bb = sh[threadIdx.x + 2 * d];
            sh[threadIdx.x] = H::add(a, bb);
        }
        __syncthreads();
    }
    if (lt == 0) sh[threadIdx.x].to_packed(reinterpret_cast<uint32_t*>(partials + (size_t)blockIdx.x * sizeof(XYZZ<F>)));
}
template <class F>
void launch_bucket_reduce_pair(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                               uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials, const uint32_t* offset, const uint32_t* count,
                               uint32_t gb0, uint32_t NB) {
    static_assert(2 * sizeof(XYZZH<typename F::Params>) == sizeof(BU<F>), "a lane pair holds one bucket");
    k_bucket_reduce_pair<F><<<nparts, rblock * 2, (size_t)rblock * sizeof(BU<F>), st>>>(buckets, half, L, lanes_per_window, wgs_per_window, partials, offset,
                                                                                       count, gb0, NB);
}
#define MSM_INSTANTIATE_REDUCE_PAIR(F)                                                                                                       \
    template void launch_bucket_reduce_pair<F>(hipStream_t, uint32_t, uint32_t, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t*, \
                                               const uint32_t*, const uint32_t*, uint32_t, uint32_t);

// The accumulation kernel proper and the rest (tail kernels, reduction) are instantiated in different translation units:
// the former is compiled with MSM_U_CHAINED_MADS (field_u.hpp), the latter -- kernels that run few waves -- without.
template <class F, int VARIANT>
void launch_accumulate(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                       const uint32_t* order, uint32_t nb, uint8_t* buckets, uint32_t seg, uint32_t seg_t, void* ctl, void* work, void* biglist) {
    constexpr uint32_t T = UKind<F>::type::LANES;
#if defined(MSM_U_CHAINED_MADS)
    static_assert(VARIANT == 1, "this translation unit builds the chained variant");
#else
    static_assert(VARIANT == 0, "this translation unit builds the two-chain variant");
#endif
    // one wave per workgroup: a SIMD takes the next wave as soon as one of its own finishes instead of waiting until a whole
    // CU can host four (measured: 2.29 -> 2.24 ms for the kernel at n = 2^20)
#ifndef MSM_ACC_BLOCK
#define MSM_ACC_BLOCK 64
#endif
#if MSM_ACC_STAGED
    static_assert(MSM_ACC_BLOCK == 64, "the staged entry lists take single-wave workgroups");
#endif
    k_accumulate<F, VARIANT><<<(uint32_t)(((uint64_t)nb * T + MSM_ACC_BLOCK - 1) / MSM_ACC_BLOCK), MSM_ACC_BLOCK, 0, st>>>(bases, xbeta, sorted, offset, count, order, nb, buckets, seg, seg_t, (TailCtl*)ctl,
                                                                                (uint4*)work, (uint4*)biglist);
}
template <class F, int VARIANT>
void launch_accumulate_slices(hipStream_t st, uint32_t k, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset,
                              const uint32_t* count, const uint32_t* order, uint32_t nb, uint8_t* slices, uint32_t seg) {
#if defined(MSM_U_CHAINED_MADS)
    static_assert(VARIANT == 1, "this translation unit builds the chained variant");
#else
    static_assert(VARIANT == 0, "this translation unit builds the two-chain variant");
#endif
    constexpr uint32_t PER_WAVE = 64u / UKind<F>::type::LANES;
    k_accumulate_slices<F, VARIANT><<<((nb + PER_WAVE - 1u) / PER_WAVE) * k, 64, 0, st>>>(bases, xbeta, sorted, offset, count, order, nb, k, slices, seg);
}
template <class F>
void launch_join_slices(hipStream_t st, uint32_t k, const uint32_t* count, const uint32_t* order, uint32_t nb, const uint8_t* slices, uint8_t* buckets,
                        uint32_t seg, void* ctl, void* work, void* biglist) {
    constexpr uint32_t T = UKind<F>::type::LANES;
    if (k >= 8 && (k & (k - 1)) == 0 && k * T <= 256) {
        const uint32_t B = 256u / (k * T);
        k_join_slices_tree<F><<<(nb + B - 1u) / B, 256, (size_t)256 * sizeof(BU<F>), st>>>(count, order, nb, k, slices, buckets, seg, (TailCtl*)ctl, (uint4*)work,
                                                                                      (uint4*)biglist);
        return;
    }
    k_join_slices<F><<<(uint32_t)(((uint64_t)nb * T + 63u) / 64u), 64, 0, st>>>(count, order, nb, k, slices, buckets, seg, (TailCtl*)ctl, (uint4*)work, (uint4*)biglist);
}
template <class F, int VARIANT>
void launch_accumulate_items(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                             const uint32_t* order, const uint32_t* itab, uint32_t T, uint32_t osh, uint32_t max_items, uint8_t* buckets, uint8_t* items_out) {
#if defined(MSM_U_CHAINED_MADS)
    static_assert(VARIANT == 1, "this translation unit builds the chained variant");
#else
    static_assert(VARIANT == 0, "this translation unit builds the two-chain variant");
#endif
    constexpr uint32_t PER_WAVE = 64u / UKind<F>::type::LANES;
    k_accumulate_items<F, VARIANT><<<(max_items + PER_WAVE - 1u) / PER_WAVE, 64, 0, st>>>(bases, xbeta, sorted, offset, count, order, itab, T, osh, buckets, items_out);
}
template <class F>
void launch_join_items(hipStream_t st, const uint32_t* count, const uint32_t* order, const uint32_t* itab, uint32_t T, uint32_t osh, uint32_t max_multi,
                       const uint8_t* items_out, uint8_t* buckets, uint32_t seg, uint32_t seg_t, void* ctl, void* work, void* biglist) {
    constexpr uint32_t TL = UKind<F>::type::LANES;
    k_join_items<F><<<(uint32_t)(((uint64_t)max_multi * TL + 63u) / 64u), 64, 0, st>>>(count, order, itab, T, osh, items_out, buckets, seg, seg_t, (TailCtl*)ctl, (uint4*)work,
                                                                                    (uint4*)biglist);
}
template <class F>
void launch_accumulate_tail(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset,
                            const uint32_t* count, uint8_t* buckets, uint32_t seg, void* ctl, void* work, void* biglist, uint8_t* seg_out,
                            uint32_t tail_block, bool tail_expected) {
    // All three loop over whatever was queued with any grid.  When nothing is expected (random scalars practically never overflow a
    // segment) the grids are small: a launch ends when its LAST workgroup has found the queue empty, and on a chip whose SIMDs are full
    // of other calls' accumulation waves every workgroup waits for register room -- the empty 1024-workgroup launch took 55 us there.
    const uint32_t g_acc = tail_expected ? 1024u : 16u, g_blk = tail_expected ? 64u : 4u, g_lane = tail_expected ? 512u : 8u;
    k_accumulate_tail<F><<<g_acc, 256, 0, st>>>(bases, xbeta, sorted, offset, count, (const TailCtl*)ctl, (const uint4*)work, seg_out);
    (void)seg;
    constexpr uint32_t T = UKind<F>::type::LANES;   // tail_block and rblock count LOGICAL lanes (points); a point takes T threads
    k_combine_tail<F><<<g_blk, tail_block * T, (size_t)tail_block * T * sizeof(BU<F>), st>>>((const TailCtl*)ctl, (const uint4*)biglist, seg_out, buckets);
    k_combine_tail_lanes<F><<<g_lane, 64, 0, st>>>((const TailCtl*)ctl, (const uint4*)biglist, seg_out, buckets);
}
template <class F>
void launch_bucket_reduce(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                          uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials, const uint32_t* offset, const uint32_t* count,
                          uint32_t gb0, uint32_t NB) {
    constexpr uint32_t T = UKind<F>::type::LANES;
    k_bucket_reduce<F><<<nparts, rblock * T, (size_t)rblock * T * sizeof(BU<F>), st>>>(buckets, half, L, lanes_per_window, wgs_per_window, partials, offset,
                                                                                         count, gb0, NB);
}

// bytes of one bucket in HBM (the engine sizes its bucket arrays and LDS tiles with this)
template <class F>
size_t bucket_bytes() {
    return sizeof(BU<F>) * UKind<F>::type::LANES;
}
// threads that share one point (2 for the lane-paired Fp2 kind)
template <class F>
uint32_t lanes_per_point() {
    return UKind<F>::type::LANES;
}

#define MSM_INSTANTIATE_ACCUMULATE(F, VARIANT)                                                                                               \
    template void launch_accumulate<F, VARIANT>(hipStream_t, const uint8_t*, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, \
                                       uint32_t, uint8_t*, uint32_t, uint32_t, void*, void*, void*);
#define MSM_INSTANTIATE_SLICES(F, VARIANT)                                                                                                   \
    template void launch_accumulate_slices<F, VARIANT>(hipStream_t, uint32_t, const uint8_t*, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, \
                                                       const uint32_t*, uint32_t, uint8_t*, uint32_t);
#define MSM_INSTANTIATE_JOIN(F)                                                                                                              \
    template void launch_join_slices<F>(hipStream_t, uint32_t, const uint32_t*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, uint32_t, void*, void*, void*);
#define MSM_INSTANTIATE_ITEMS(F, VARIANT)                                                                                                    \
    template void launch_accumulate_items<F, VARIANT>(hipStream_t, const uint8_t*, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*, \
                                                      const uint32_t*, uint32_t, uint32_t, uint32_t, uint8_t*, uint8_t*);
#define MSM_INSTANTIATE_JOIN_ITEMS(F)                                                                                                        \
    template void launch_join_items<F>(hipStream_t, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t, uint32_t, const uint8_t*, uint8_t*, uint32_t, \
                                       uint32_t, void*, void*, void*);
#define MSM_INSTANTIATE_TAIL(F)                                                                                                              \
    template void launch_accumulate_tail<F>(hipStream_t, const uint8_t*, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, uint8_t*,    \
                                            uint32_t, void*, void*, void*, uint8_t*, uint32_t, bool);
#define MSM_INSTANTIATE_REDUCE(F)                                                                                                            \
    template void launch_bucket_reduce<F>(hipStream_t, uint32_t, uint32_t, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t*, const uint32_t*, \
                                          const uint32_t*, uint32_t, uint32_t); \
    template void launch_bucket_planes<F>(hipStream_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint8_t*, uint8_t*, uint8_t*, const uint32_t*, \
                                          const uint32_t*, uint32_t, uint32_t); \
    template uint32_t planes_lane_bits<F>();                                                                                                  \
    template size_t bucket_bytes<F>();                                                                                                        \
    template uint32_t lanes_per_point<F>();
#define MSM_INSTANTIATE_TAIL_AND_REDUCE(F) MSM_INSTANTIATE_TAIL(F) MSM_INSTANTIATE_REDUCE(F)
#define MSM_INSTANTIATE_BUCKET_KERNELS(F) MSM_INSTANTIATE_ACCUMULATE(F, 0) MSM_INSTANTIATE_TAIL_AND_REDUCE(F)

}  // namespace msm
