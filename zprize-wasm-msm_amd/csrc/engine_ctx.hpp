// engine_ctx.hpp -- the host-side state behind msm_amd_ctx (slots, streams, scratch buffers), error reporting and the small host
// helpers that every translation unit of the library shares (engine.hip, conv_kernels.hip, fft_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <map>
#include <atomic>
#include <functional>
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
// A scratch buffer in HBM that grows on demand (ensure()).  `gen` counts its allocations: whatever the engine remembers ABOUT THE
// CONTENTS of a buffer ("all zero since the last call") is keyed on (pointer, generation), never on the pointer alone -- a buffer
// that grows is freed and allocated anew, and the allocator may hand the same address back with the larger size (round 2: the sort's
// bin histogram was then taken for still-zero beyond its old capacity, one GPU memory fault).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    uint64_t gen = 0;   // 0 = never allocated
};
// what ensure() allocates with; a host-only test swaps in an allocator that returns the same address again (tests/host/test_devbuf.hip)
struct DevAllocHooks {
    hipError_t (*alloc)(void**, size_t) = [](void** p, size_t n) { return hipMalloc(p, n); };
    hipError_t (*release)(void*) = [](void* p) { return hipFree(p); };
};
inline DevAllocHooks& dev_alloc_hooks() {
    static DevAllocHooks h;
    return h;
}
// "this buffer has held nothing but zeros since allocation `gen`" -- a mark taken with mark_zero() and tested with is_zero()
struct ZeroMark {
    const void* p = nullptr;
    uint64_t gen = 0;
    void set(const DevBuf& b) { p = b.p; gen = b.gen; }
    void clear() { p = nullptr; gen = 0; }
    bool holds(const DevBuf& b) const { return b.p != nullptr && p == b.p && gen == b.gen; }
};

}  // namespace msm

using namespace msm;

// One submission as the entry points describe it (engine.hip) and the call's stages read it (engine_call.hpp).
struct RunArgs {
    const void* bases;
    const void* scalars;
    uint32_t scalar_size, n, flags;
    uint32_t bit_lo, nbits;   // scalar bit range
    uint32_t shard, nshards;
    bool raw_out;             // un-normalised Jacobian partial (multi-GPU shard)
    msm_amd_bases* prepared = nullptr;   // bases = the handle's window table (msm_amd_multiexp_prepared)
    bool full_width = false;             // the retry of a call whose narrow layout met a wider scalar: tile the declared width, ignore the hints
};

// One in-flight MSM: its scratch buffers, events and what the host needs to finish it.  A context owns
// two slots so that the bucket reduction of one call (latency-bound, one wave per SIMD, stream B) overlaps
// the sort + accumulation of the next call (throughput-bound, stream A).
struct Slot {
    DevBuf d_itab;    // the work list's two class prefix tables (k_items_plan)
    DevBuf d_planes;  // bit-plane reduction: what the workgroups of a window leave for k_planes_combine
    int plane_shift = -1;   // >= 0: the partials of a window are bit planes (finish_msm: partial 1+b weighs 2^(b + plane_shift)); -1: equal weights
    DevBuf d_halves, d_tmp, d_binhist, d_binstart, d_bincur, d_cpre, d_desc, d_bcursor, d_tailctl, d_work, d_biglist, d_segout, d_slices;
    DevBuf d_xbeta;   // GLV path: beta * x of every base
    DevBuf d_bases, d_scalars, d_affine, d_count, d_offset, d_order, d_bins, d_sorted, d_buckets, d_partials;
    ZeroMark binhist_zero;        // holds(d_binhist) while that buffer is known to be all zero (sort: k_bin_scan leaves it so)
    ZeroMark ctl_zero, flags_zero;   // the same for the first 16 bytes of d_tailctl and d_flags (k_collect_tail leaves them so)
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
    bool narrow_layout = false;     // the windows tile a width narrower than the declared one (the context's narrow hint): scalars beyond them make the collector run the call again
    bool top_is_overflow = false;   // the top owned window only holds what lies beyond the expected scalar width (plan(): empty for scalars below the group order)
    msm_amd_bases* prepared = nullptr;   // the call runs on a prepared table (released from in_flight when collected)
    DevBuf d_flags;                      // word 0: a scalar had bits beyond the windows of the prepared table
    int (*finish)(msm_amd_ctx*, Slot*, uint8_t*) = nullptr;
    int deferred_rc = 0;                        // what queueing the deferred reduction returned, for the wait on the ticket
    std::function<int(bool)> deferred_reduce;   // the call's reduction, not queued yet (argument: are other calls in flight now?); under ctx->mu
    RunArgs retry_args{};            // what was submitted, kept so that the collector can run the call again (narrow layout met a wide scalar)
    int retry_curve = 0, retry_group = 0;
    DevBuf* bufs[27] = {&d_planes, &d_itab, &d_flags, &d_slices, &d_xbeta, &d_halves, &d_tmp, &d_binhist, &d_binstart, &d_bincur, &d_cpre, &d_desc, &d_bcursor, &d_tailctl, &d_work, &d_biglist, &d_segout, &d_bases, &d_scalars,
                        &d_affine, &d_count, &d_offset, &d_order, &d_bins, &d_sorted, &d_buckets, &d_partials};
};

constexpr uint32_t MAX_SLICES = 64;   // lanes per bucket of a sliced accumulation

struct msm_nccl_id {   // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
    char internal[128];
};

// Prepared bases (msm_amd_prepare_bases): the window table of a fixed set of points, resident in HBM.
struct msm_amd_bases {
    msm_amd_ctx* ctx = nullptr;
    int curve = 0, group = 0;
    uint32_t n = 0;             // points
    uint32_t stride = 0;        // points per table row: n rounded up to the sort's largest tile (8192), so that a tile of scalars never straddles an index slice
    uint32_t scalar_bits = 0;   // the windows cover scalars below 2^scalar_bits
    uint32_t c = 0, W = 0, ns = 0;   // window layout (Geom: ns windows of c bits, signed, then W - ns of c - 1 bits, unsigned)
    void* table = nullptr;      // [W][stride] affine points, the reference's packed Montgomery image: table[w][i] = 2^(start bit of window w) * P_i
    size_t table_bytes = 0;
    std::atomic<int> in_flight{0};   // calls submitted and not yet collected (release refuses while > 0)
};

inline void msm_bases_kind(const msm_amd_bases* h, int* curve, int* group) {
    *curve = h->curve;
    *group = h->group;
}

struct msm_amd_ctx {
    int device = 0;
    // multi-GPU group this context belongs to (dist.hip): RCCL communicator + staging for the exchange of the window partials
    // (guarded by dist_mu)
    struct Dist {
        void* comm = nullptr;
        int rank = 0, world = 0;
        void* d_buf = nullptr;
        void* h_buf = nullptr;   // pinned
        hipStream_t stream = nullptr;
        size_t slot_bytes = 0;
        // One record per call of the group that this rank has submitted and not yet exchanged.  The exchanges (collectives of one
        // communicator) happen in submission order -- `seq` -- on every rank, whichever thread waits for which ticket when.
        struct Rec {
            uint64_t seq = 0, engine_ticket = 0;
            int curve = 0, group = 0;
            int rc = 0;          // != 0: this rank's part could not even be queued; it still takes part in the exchange, with a failure mark
            std::string err;
            bool waiting = false;
            // what this rank believes about the call, sent in the exchange's header and compared across ranks (a disagreement -- options
            // that differ between ranks -- would otherwise add partials of different splits into a silently wrong point):
            // gp | gw << 16, the total number of points (0: prepared tables, whose slices need not tile anything), a hash of the window
            // layout of this rank's part (0: nothing to compare -- an empty part)
            uint32_t split = 0, n_total = 0, layout = 0;
            // Exchanges happen in submission order, tickets may be waited for in ANY order: a thread that waits for a later ticket performs
            // the exchanges of the earlier ones on the way and parks their outcome here for whoever asks for them
            bool collected = false;   // this rank's own partial has been taken from the engine (its slot is free)
            bool done = false;        // the exchange of this call has happened: out_rc / out_err / out are final
            int out_rc = 0;
            std::string out_err;
            std::vector<uint8_t> out;   // the normalised sum (3 field elements)
        };
        std::map<uint64_t, Rec> recs;   // by dist ticket
        uint64_t next_id = 1, next_seq = 0, next_exchange = 0;
        bool exchanging = false;   // a thread has taken its turn and is inside the collective -- OUTSIDE dist_mu; staging, stream and comm are its alone
        // A group is DEAD once an exchange has failed locally or timed out (a peer that never entered): the communicator has been
        // aborted (ncclCommAbort), every pending and every later call of the group returns MSM_AMD_EPEER at once instead of blocking,
        // and the only way on is msm_amd_dist_destroy (+ a new msm_amd_dist_init).
        bool dead = false, comm_gone = false;   // comm_gone: the communicator was aborted -- nothing left to destroy
        std::string dead_why;
    } dist;
    mutable std::mutex dist_mu;
    std::condition_variable dist_cv;   // next_exchange moved on, an exchange ended, the group died or was destroyed
    uint64_t dist_epoch = 0;           // bumped by dist_init / dist_destroy (under dist_mu): a waiter that slept through a destroy + init must not take the new group for its own
    std::atomic<uint32_t> dist_timeout_ms{300000};   // option "dist_timeout_ms" / MSM_AMD_DIST_TIMEOUT_MS: how long an exchange may wait for the peers (0 = for ever)
    std::vector<msm_amd_bases*> prepared_handles;   // handles this context has prepared and not released (guarded by mu): freed by msm_amd_destroy
    int even_windows = 1;              // option "even_windows": windows of c and c - 1 bits that tile the scalar exactly when c does not divide its width
    uint32_t dist_gw = 0;              // option "dist_gw": window groups of a sharded call (0 = chosen per call; must divide the world size)
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
    // what the last finished call put into its buckets: its bucket count and 16 x the mean entries per bucket.  A call with the same bucket
    // count takes it as the expected density of its own input (reduce(): sparse inputs -- a witness full of zeros, ones and small values --
    // have a short accumulation that cannot hide a long reduction)
    std::atomic<uint32_t> fill_nb{0}, fill_mean16{0};
    // option "defer_reduce" (1): the reduction of a call that takes part in a stream of calls is queued when the next thing happens on the
    // context -- the next submission (it will hide behind that call's accumulation: the form for that) or the wait for its own ticket (nothing
    // followed: the lone form, less than half as long) -- instead of at its own submission, where the engine can only guess (engine_call.hpp)
    int defer_reduce = 1;
    int burst_hint = 0;      // > 0: calls were submitted with others in flight lately (8 at such a submission, one less at every other one)
    int reduce_planes = 1;   // option "reduce_planes": the bucket reduction as plain sums (k_bucket_planes): 1 = where it pays (reduce()), 2 = wherever it can, 0 = k_bucket_reduce always
    uint32_t force_L = 0;   // MSM_AMD_REDUCE_L: buckets per lane in k_bucket_reduce (tuning)
    int no_hot = 0;         // option "hot" = 0: no short segments / grouped fold for hot buckets
    int zero_copy = 1;      // option "zero_copy" (default 1): the reduction writes its partials and the tail words straight into the pinned host buffer -- no copy back (15 us of every call); 0 = device buffer + one copy
    int no_split2 = 0;      // option "split2" = 0: one lane per bucket always (switches k_accumulate_slices off)
    uint32_t acc_variant = 0, acc_stream = 0, tile_mode = 0, chunk_mode = 0, sort_geom = 0;
    int items_mode = 0;     // option "items": the accumulation's work list (0 = by launch size, 1 = wherever the populations allow, 2 = never)
    uint32_t items_t = 0;   // option "items_t": entries per lane of the work list (0 = chosen per launch)
    uint32_t splitk = 0;    // option "splitk": lanes per bucket of the G1 accumulation, 0 = chosen per launch, 1..MAX_SLICES forced
    uint32_t splitk_lanes = 5u * 1024u * 64u;   // option "splitk_lanes": the lane count a launch with few buckets is brought to (five waves for each of 1024 SIMDs)
    uint32_t max_slice = 1u << 26;   // synchronous calls with more points are cut into slices of this many (option "slice")
    int glv_mode = 1;       // option "glv": 0 = never, 1 = where it pays (BLS12-381 G1, n <= 2^17), 2 = whenever it applies
    int no_split = 1;       // option "split" = 1 splits a lone call into two window groups (measured slower: off by default)
    int reduce_pair = 0;    // option "reduce_pair" = 1: G1 bucket reduction with every point on a lane pair (measured: no gain)
    std::atomic<int> wide_hint{0};   // > 0: recent calls had scalars beyond the group order's bit length (entries in the overflow window): windows tile the declared width
    // observed scalar width (engine_call.hpp plan()): the widest scalar of each of the last eight eligible calls; narrow_bits = their
    // maximum once two have been seen (0 = nothing known): the width the next calls' windows tile when it is below the group order's
    std::mutex hint_mu;
    uint16_t width_hist[8] = {};
    uint32_t width_count = 0;
    std::atomic<int> narrow_bits{0};
    int narrow_mode = 1;    // option "narrow": 0 = never adapt the windows to the observed scalar width
    std::atomic<int> skew_hint{0};   // > 0: recent calls had buckets beyond the segment length; the sort then expects long spans
    int sort_big = 0;       // option "sort_big": 0 = chunk-parallel sort of long spans where they are expected, 1 = always, 2 = never (tests)
    std::atomic<size_t> scratch_level[2] = {};   // bytes per lane of private segment that streams A / S are known to hold scratch for (msm_scratch_guard)
    uint32_t last_items_T = 0;   // entries per lane of the work list in the call planned last (0: it ran one lane per bucket or slices)
    float timings[8] = {};
    uint32_t geometry[6] = {};
};

// curve and group of a prepared table, looked up among the context's registered tables before it is dereferenced (false: not a table
// of this context, or released)
inline bool msm_bases_lookup(msm_amd_ctx* ctx, const msm_amd_bases* h, int* curve, int* group) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (std::find(ctx->prepared_handles.begin(), ctx->prepared_handles.end(), h) == ctx->prepared_handles.end()) return false;
    *curve = h->curve;
    *group = h->group;
    return true;
}

// ---- scratch (private segments) ------------------------------------------------------------------------------------------------
// A kernel that needs P bytes of private segment per lane makes the HSA runtime give the hardware queue it is dispatched on
// P x 64 lanes x 8192 wave slots of scratch (measured: tools/microbench/scratch_dispatch.hip, profiles/r04_scratch_dispatch.txt), at
// the dispatch, kept from then on, grown when a later kernel needs more.  When that allocation fails the runtime ABORTS THE PROCESS --
// nothing a library can catch.  So: the kernels of the MSM path carry at most MSM_SCRATCH_RESERVE bytes (spills; their additions are
// inlined: bucket_kernels.hpp; the build fails otherwise: tools/check_code_objects.py), and msm_amd_create dispatches a kernel with
// exactly that much on every stream of the context after checking that the memory is there -- a device too full for it fails the
// CREATE with MSM_AMD_ENOMEM.  The other entry points (group FFTs, batch conversions, the table build: 0.3-2.7 KB per lane, out-of-line
// point arithmetic) ask msm_scratch_guard before they launch: it reads the kernel's private segment size from the code object
// (hipFuncGetAttributes) and returns MSM_AMD_ENOMEM when the device could not take the queue's scratch plus a margin.
constexpr size_t MSM_SCRATCH_RESERVE = 128;                    // bytes per lane reserved at create on every stream
constexpr size_t MSM_SCRATCH_SLOTS = (size_t)64 * 8192;        // lanes the runtime sizes a queue's scratch for
constexpr size_t MSM_SCRATCH_MARGIN = (size_t)64 << 20;
// slot: 0 = stream A (conversions, FFTs), 1 = stream S (table build)
int msm_scratch_guard(msm_amd_ctx* ctx, int slot, const void* kernel, const char* what);
template <class K>
inline int msm_guard(msm_amd_ctx* ctx, int slot, K* kernel, const char* what) {
    return msm_scratch_guard(ctx, slot, reinterpret_cast<const void*>(kernel), what);
}

// Error text of a context.  Written by whichever thread fails (several may use one context: the libuv workers of the Node addon),
// so it is guarded by its own small lock, and msm_amd_last_error hands out a per-thread copy.
void msm_set_err(msm_amd_ctx* ctx, const std::string& msg);

// Window layout of the call behind an engine ticket that has been submitted and not yet collected (engine.hip); false: no such
// ticket, or a call without GPU work (n = 0)
bool msm_ticket_geom(msm_amd_ctx* ctx, uint64_t ticket, Geom* g);
// The split of a call of n points over `world` ranks (dist.hip): gp point slices x gw window groups, gp * gw = world.
void msm_dist_split(const msm_amd_ctx* ctx, int curve, int group, uint32_t scalar_size, uint32_t n, uint32_t world, uint32_t* gp, uint32_t* gw);
// Slice p of gp: points [first, first + count) -- equal slices of ceil(n / gp), the last ones shorter or empty.
inline void msm_dist_slice(uint32_t n, uint32_t gp, uint32_t p, uint32_t* first, uint32_t* count) {
    const uint64_t per = ((uint64_t)n + gp - 1) / gp, lo = std::min<uint64_t>((uint64_t)p * per, n), hi = std::min<uint64_t>(lo + per, n);
    *first = (uint32_t)lo;
    *count = (uint32_t)(hi - lo);
}

#define HIPCHK(ctx, call)                                                                                      \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            msm_set_err((ctx), std::string(#call) + ": " + hipGetErrorString(e_));                             \
            (void)hipGetLastError(); /* reported: the runtime's sticky copy must not fail the NEXT call's check */ \
            return e_ == hipErrorOutOfMemory ? MSM_AMD_ENOMEM : MSM_AMD_EHIP;                                  \
        }                                                                                                      \
    } while (0)

// a failed device allocation outside HIPCHK: MSM_AMD_ENOMEM, the runtime's sticky error cleared
inline int msm_alloc_failed(msm_amd_ctx* ctx, const char* what) {
    (void)hipGetLastError();
    msm_set_err(ctx, std::string(what) + ": out of device memory");
    return MSM_AMD_ENOMEM;
}

inline int ensure(msm_amd_ctx* ctx, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes && b.p) return 0;
    if (b.p) HIPCHK(ctx, dev_alloc_hooks().release(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, dev_alloc_hooks().alloc(&b.p, want));
    b.cap = want;
    ++b.gen;   // a new allocation: nothing known about its contents, whatever its address
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

template <class P>
static Fp<P> fp_from(const uint32_t* limbs) {
    Fp<P> r;
    for (int i = 0; i < P::N; ++i) r.v[i] = limbs[i];
    return r;
}

// k_jacobian_to_affine: g?m_batchToAffine and the Jacobian-bases entry of the MSM share it
namespace msm {
// ------------------------------------------------------------------------------------------
// Jacobian bases -> affine (g?m_multiexp entry point): Montgomery's trick, 8 points per lane
// (build_curve_jacobian_a0.js:1040-1125 batchToAffine does the same thing sequentially).
// ------------------------------------------------------------------------------------------
// (every loop over the lane's points is fully unrolled, so that the prefix products stay in registers: indexed dynamically the array lived
// in the private segment -- 400-784 bytes per lane, which the HSA runtime turns into up to 0.4 GB of scratch per hardware queue)
template <class F>
__global__ void __launch_bounds__(256) k_jacobian_to_affine(const uint8_t* __restrict__ jac, uint32_t n, uint8_t* __restrict__ aff) {
    constexpr int K = sizeof(F) > 48 ? 4 : 8;   // points per lane (Fp2 elements are twice as wide: half as many keep the registers in bounds)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t first = t * K;
    if (first >= n) return;
    F pref[K];
    F acc = F::one();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        uint32_t i = first + k;
        pref[k] = acc;
        if (i < n) {
            F z = load16<F>(jac + (size_t)i * 3 * sizeof(F) + 2 * sizeof(F));
            if (!z.is_zero()) acc = F::mul(acc, z);
        }
    }
    F inv = F::inv(acc);
#pragma unroll
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
template <class F>
constexpr uint32_t jacobian_to_affine_points_per_lane() {
    return sizeof(F) > 48 ? 4u : 8u;
}

}  // namespace msm
