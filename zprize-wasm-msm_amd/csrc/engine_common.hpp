// engine_common.hpp -- helpers shared by the engine's translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "curve.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
template <class T>
MSM_D T load16(const void* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte granules");
    T t;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&t);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); ++k) {
        uint4 q = s[k];
        d[4 * k + 0] = q.x;
        d[4 * k + 1] = q.y;
        d[4 * k + 2] = q.z;
        d[4 * k + 3] = q.w;
    }
    return t;
}
template <class T>
MSM_D void store16(void* p, const T& t) {
    static_assert(sizeof(T) % 16 == 0, "16-byte granules");
    uint4* s = reinterpret_cast<uint4*>(p);
    const uint32_t* d = reinterpret_cast<const uint32_t*>(&t);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); ++k) s[k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
}

// Hot buckets (bucket_kernels.hpp): a bucket that holds more than HOT_FACTOR x the segment length -- digit 1 of window 0 of a witness
// full of ones, the carry out of the top used window of 64-bit values: 0.3 n .. 0.5 n entries -- is cut into SHORT segments (`seg_t`
// entries, chosen by the engine so that the segment lists stay bounded), its main lane adds only that many itself, and its segment sums
// are folded by HOT_GROUPS workgroups (then one): three short dependent chains instead of one of hundreds of additions.  Up to HOT_MAX
// such buckets per accumulation launch; further ones take the ordinary path (correct, slower).  Behind the tail counters in `ctl`:
// the list of hot buckets' entries in the big-bucket list, then the HOT_MAX x HOT_GROUPS partial sums.
constexpr uint32_t HOT_FACTOR = 32, HOT_GROUPS = 32, HOT_MAX = 1024, HOT_MIN_SEGS = 4096;   // (fewer segments than that: one workgroup folds them -- 32 trees over a few hundred values cost more than they save)
constexpr size_t HOT_LIST_OFFSET = 16, HOT_PART_OFFSET = HOT_LIST_OFFSET + (size_t)HOT_MAX * 4;   // bytes from `ctl`

// The work list of the accumulation (bucket_kernels.hpp k_accumulate_items): lanes per bucket by population class.
constexpr uint32_t ITEMS_NI_MAX = 8, ITEMS_CLASSES = 1024;   // (= ORDER_BINS of sort_kernels.hpp, static_asserted there)
// lanes per bucket of population class `key` (key = 1023 - min(cnt >> osh, 1023): sort_kernels.hpp order_key)
MSM_HD uint32_t items_ni(uint32_t key, uint32_t T, uint32_t osh) {
    const uint32_t pop_hi = (((ITEMS_CLASSES - 1u) - key + 1u) << osh) - 1u;   // the largest population of the class (class 0 is open-ended: its lanes cover ni * T)
    const uint32_t ni = (pop_hi + T - 1u) / T;
    return ni < 1u ? 1u : (ni > ITEMS_NI_MAX ? ITEMS_NI_MAX : ni);
}

constexpr int MAX_WINDOW_BITS = 21;   // widest window: sign + 20 bits of bucket index (8-9 coarse + 11-12 fine: the sort's two levels; the partition packs the index beside a 12-bit rank, sort_kernels.hpp)

// Window geometry shared by host and kernels.
struct Geom {
    uint32_t n;
    uint32_t scalar_size;   // bytes per scalar
    uint32_t bit_lo;        // first bit of the scalar range
    uint32_t nbits;         // bits in the range (after clipping to the scalar end)
    uint32_t c;             // window bits
    uint32_t W;             // number of windows = nbits / c + 1 (the last one absorbs the carry)
    uint32_t half;          // buckets per window = 2^(c-1)
    uint32_t shard, nshards;   // this call owns windows w with w % nshards == shard
    uint32_t Wl;            // number of owned windows
    // Prepared bases (msm_amd_prepare_bases: a table of 2^(start bit of window w) * P_i for every window w): all windows share ONE
    // bucket set (`merged`; Wl = 1), and entry (i, w) addresses table point w * stride + i.  The windows are then of two kinds, both
    // filling the same 2^(c-1) buckets evenly: the first `ns` are c bits wide and signed (digit in [-2^(c-1), 2^(c-1)], carry out),
    // the rest c - 1 bits wide and unsigned (digit + carry in [0, 2^(c-1)], no carry out -- so the top window never spills into a
    // further one).  Calls on caller-supplied bases have merged = 0 and ns = W: every window c bits, signed.
    uint32_t merged, ns, stride;
    uint32_t check_wide;    // 1: the windows cover less than the declared scalar width (a prepared table; the narrow layout of engine_call.hpp): scalar bits beyond them are REPORTED (flag word 0), never dropped silently
};
// first scalar bit of window w
MSM_HD uint32_t window_start_bit(const Geom& g, uint32_t w) { return w <= g.ns ? w * g.c : g.ns * g.c + (w - g.ns) * (g.c - 1u); }


}  // namespace msm
