// engine_call.hpp -- ONE submission of the engine, stage by stage (included by engine.hip only; split out of it in round 4).
//   pick_window_bits   the window size by n, curve, group and rank count
//   MsmCall<F>         plan() -- inputs on the GPU, window layout (windows of even width, overflow window, prepared tables, GLV), the
//                      work list's parameters; sort() -- stages 1-3 on the sort stream (csrc/sort_kernels.hpp: partition geometry by size);
//                      accumulate() -- stage 4 (bucket_kernels.hpp: one lane per bucket, the work list, slices, tail kernels);
//                      reduce() -- stage 5 and the partials' way to the host
//   finish_msm<F>      stage 6 on the host: window sums, Horner combine, normalisation, the hints for the context's next calls
// What the reference does here in one loop nest: wasmcurves/src/build_multiexp.js:251-371 (schedule), :96-249 (one window).
#pragma once
#include <memory>
#include "engine_ctx.hpp"
#include "sort_kernels.hpp"

constexpr int MSM_RETRY_FULL_WIDTH = 1;   // finish_msm -> finish_ticket: run the call again over the declared scalar width (never leaves the library)

// (defined in engine.hip, used by plan(): a stream created late gets its scratch like the others)
__global__ void k_scratch_reserve(uint32_t* out, uint32_t n, uint32_t sel);

namespace msm {


// launchers of the bucket kernels, instantiated in bucket_g1_bls12381.hip / bucket_g1_bn254.hip / bucket_g2_*.hip
template <class F, int VARIANT>
void launch_accumulate(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                       const uint32_t* order, uint32_t nb, uint8_t* buckets, uint32_t seg, uint32_t seg_t, void* ctl, void* work, void* biglist);
template <class F, int VARIANT>
void launch_accumulate_slices(hipStream_t st, uint32_t k, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset,
                              const uint32_t* count, const uint32_t* order, uint32_t nb, uint8_t* slices, uint32_t seg);
template <class F>
void launch_join_slices(hipStream_t st, uint32_t k, const uint32_t* count, const uint32_t* order, uint32_t nb, const uint8_t* slices, uint8_t* buckets,
                        uint32_t seg, void* ctl, void* work, void* biglist);
template <class F, int VARIANT>
void launch_accumulate_items(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                             const uint32_t* order, const uint32_t* itab, uint32_t T, uint32_t osh, uint32_t max_items, uint8_t* buckets, uint8_t* items_out);
template <class F>
void launch_join_items(hipStream_t st, const uint32_t* count, const uint32_t* order, const uint32_t* itab, uint32_t T, uint32_t osh, uint32_t max_multi,
                       const uint8_t* items_out, uint8_t* buckets, uint32_t seg, uint32_t seg_t, void* ctl, void* work, void* biglist);
template <class F>
void launch_accumulate_tail(hipStream_t st, const uint8_t* bases, const uint8_t* xbeta, const uint32_t* sorted, const uint32_t* offset,
                            const uint32_t* count, uint8_t* buckets, uint32_t seg, void* ctl, void* work, void* biglist, uint8_t* seg_out,
                            uint32_t tail_block, bool tail_expected);
template <class F>
void launch_bucket_reduce(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                          uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials, const uint32_t* offset, const uint32_t* count,
                          uint32_t gb0, uint32_t NB);
template <class F>
void launch_bucket_planes(hipStream_t st, uint32_t windows, uint32_t half, uint32_t L, uint32_t mlog, const uint8_t* buckets, uint8_t* wvout, uint8_t* partials,
                          const uint32_t* offset, const uint32_t* count, uint32_t gb0, uint32_t NB);   // the reduction as plain sums (bit planes)
template <class F>
uint32_t planes_lane_bits();   // log2 of the points one wave holds in the plane kernels
template <class F>
void launch_bucket_reduce_pair(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                               uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials, const uint32_t* offset, const uint32_t* count,
                               uint32_t gb0, uint32_t NB);   // G1 only: every point of the reduction on a lane pair (XYZZH, curve_u.hpp)
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

// bit length of the group order r: what scalars reduced mod r -- a prover's -- need of their 32 bytes
template <class F>
struct OrderBits {
    static constexpr uint32_t value = 256;
};
template <bool U>
struct OrderBits<Fp<Bls12381Fp, U>> {
    static constexpr uint32_t value = 255;
};
template <>
struct OrderBits<Fp2<Bls12381Fp>> {
    static constexpr uint32_t value = 255;
};
template <bool U>
struct OrderBits<Fp<Bn254Fp, U>> {
    static constexpr uint32_t value = 254;
};
template <>
struct OrderBits<Fp2<Bn254Fp>> {
    static constexpr uint32_t value = 254;
};

}  // namespace msm


// mid_from / mid_below: the range of n that takes 14-bit windows of even width (0, 0 = none)
static uint32_t pick_window_bits(uint32_t n, uint32_t nbits, uint32_t forced, uint32_t nshards, uint32_t narrow_below, bool even = true, uint32_t mid_from = 0,
                                 uint32_t mid_below = 0) {
    uint32_t c;
    if (forced) {
        c = forced;
    } else if (n >= (1u << 23)) {
        // From n = 2^23 the additions saved by fewer, wider windows (14 windows of 19 bits / 13 of 20 over a 255-bit scalar instead of
        // 16 of 16 bits) outweigh the 8-16 times larger bucket arrays and their reduction (profiles/r02_sweep_window_bits_large_n.txt:
        // n = 2^24 39.0 ms at c = 19 against 42.7 at c = 16, n = 2^26 145.3 ms at c = 20 against 177.3; the reference widens its window
        // with n too, build_multiexp.js:275-280).  A sharded call is as slow as its busiest rank, which owns ceil(W / nshards) windows:
        // 13 windows over 8 ranks are 2 + 1, 16 windows are 2 + 2 -- so the choice minimises
        //     (windows of the busiest rank) x (n bucket additions + 2^(c-1) buckets x the cost of reducing one, ~6 additions' worth).
        uint64_t best = ~0ull;
        c = 16;
        for (uint32_t cand : {16u, 19u, 20u}) {
            if (cand > nbits) continue;
            const uint32_t w = (nbits + cand - 1) / cand;   // windows that receive scalar bits (a further carry-only window is negligible)
            // (the bucket term: ~3 additions' worth per bucket when its reduction overlaps the next call's accumulation -- measured at
            // n = 2^23, two calls in flight: 21.20 ms at c = 16, 21.73 at 19, 20.85 at 20, profiles/r03_sweep_prepared_window_bits.txt;
            // a lone call pays ~5 per bucket and is level there, 21.83 against 21.80)
            uint64_t cost = (uint64_t)((w + nshards - 1) / nshards) * ((uint64_t)n + 3ull * (1ull << (cand - 1)));
            // a top window of fewer than 12 bits (c = 19 over 256 bits: 9) has a few hundred buckets of n / 2^tb entries each, which go
            // through the tail kernels at half the efficiency: count it twice (n = 2^24, ms per MSM at c = 19 | 20: BN254 20.8 | 20.2,
            // BLS12-381 40.5 | 37.8 -- k_accumulate_tail + k_combine_tail took 2.6 of BN254's 20.8)
            // (with windows of even width -- submit_msm -- there is no such top window)
            const uint32_t tb = nbits - (w - 1) * cand;
            if (!even && tb < 12 && tb < cand) cost += n;
            if (cost < best) {
                best = cost;
                c = cand;
            }
        }
    } else {
        // measured on MI355X (profiles/r01_sweep_window_bits*.log): the bucket reduction costs about the same whatever n is
        // (latency-bound), so 2^15 buckets per window win as soon as n reaches a few thousand; below that, 8 bits.
        // Where the accumulation cuts buckets into slices (unsharded calls) 8-bit windows stay ahead much longer: 33 x 128 buckets
        // of hundreds of entries run as up to 64 lanes each and reduce in 0.28 ms instead of 0.7 (profiles/r02_sweep_window_bits_small_n.txt,
        // ms per MSM, c = 8 | 16: BLS12-381 G1 2^12 0.42 | 1.00, 2^14 0.49 | 0.97, 2^15 0.60 | 0.99, 2^16 0.83 | 1.02, 2^17 1.11 | 1.10;
        // BN254 G1 2^14 0.30 | 0.56, 2^16 0.48 | 0.58, 2^17 0.56 | 0.62, 2^18 0.92 | 0.68; BLS12-381 G2 2^12 0.65 | 1.58, 2^14 0.90 | 1.57,
        // 2^16 1.73 | 1.75, 2^17 2.74 | 2.15; BN254 G2 2^14 0.48 | 0.89, 2^16 0.77 | 0.93, 2^17 1.31 | 1.06): narrow_below = 98304 for
        // G1, 65537 for G2.  Widths between 9 and 15 lose to both: 256 is not a multiple of them and the few dense buckets of their
        // narrow top window set the duration of the accumulation.
        // With windows of even width (submit_msm) the sizes in between are usable: 14 bits -- 19 windows of 14 / 13 bits over 2^13
        // buckets each.  ms per MSM, calls in flight as bench.py keeps them, c = 8 | 12 | 14 | 16 (profiles/r03_even_windows_and_slices.txt,
        // r03_even_windows_other_curves.txt): BLS12-381 G1 n = 2^15 .587 | .618 | .471 | -, 2^16 .790 | .740 | .599 | 1.036, 2^17 1.103 | .991 |
        // .762 | 1.105, 2^18 1.820 | 1.585 | 1.041 | 1.251 (three in flight), 2^19 - | - | 2.043 | 1.685; below 2^15, 8 bits stay ahead (2^14
        // .488 | .577 | .639 | .983).  BN254 G1 2^16 .521 | .494 | .468 | .600, 2^17 .672 | .577 | .585 | .645, 2^18 1.087 | .797 | .761 | .759.
        // BLS12-381 G2 2^16 1.707 | 1.553 | 1.401 | 1.759, 2^17 2.718 | 2.474 | 1.795 | 2.118, 2^18 4.726 | 4.657 | 3.152 | 2.918.  BN254 G2 gains
        // nothing (2^16 1.229 | - | .953 | .953, 2^18 3.223 | - | 2.056 | 1.358) and keeps the round-2 rule, as do sharded calls.
        const uint32_t wide_from = nshards == 1 ? narrow_below : 4096u;
        c = n >= wide_from ? 16 : 8;
        if (even && nshards == 1 && mid_from && n >= mid_from && n < mid_below) c = 14;
    }
    if (c > nbits) c = nbits;
    if (c < 1) c = 1;
    if (c > (uint32_t)MAX_WINDOW_BITS) c = MAX_WINDOW_BITS;   // sign + 8 coarse + 12 fine bits of bucket index (sort_kernels.hpp)
    return c;
}


template <class F>
static int finish_msm(msm_amd_ctx* ctx, Slot* sl, uint8_t* out);

// One submission of the engine, stage by stage.  The members are what the stages share: where the inputs live, the window geometry,
// the slot's tables, the streams.  plan() decides everything (window layout, GLV schedule, limits) and brings the inputs to the GPU;
// sort() queues stages 1-3 on the sort stream, accumulate() stage 4, reduce() stage 5 and the downloads of the partials.
template <class F>
struct MsmCall {
    using X = XYZZ<F>;
    using FK = typename KernelField<F>::type;
    static constexpr size_t ASZ = sizeof(Affine<F>), JSZ = sizeof(Jacobian<F>), XSZ = sizeof(X);   // packed images
    static constexpr int kNothingToQueue = 1;   // plan(): the result is infinity (n == 0, empty bit range, no owned window)
    msm_amd_ctx* ctx;
    Slot* sl;
    const RunArgs& a;
    const size_t BSZ = bucket_bytes<FK>();   // one bucket in HBM
    bool others_pending = false, use_glv = false, split_mode = false, acc_shared = false;   // (acc_shared: the accumulation runs on the context's stream A)
    hipStream_t st = nullptr;                // the stream the current stage queues on
    const uint8_t *d_bases = nullptr, *d_scalars = nullptr, *d_xbeta = nullptr;
    Geom g{};
    uint32_t nv = 0, nve = 0, NB = 0, nblk = 0, oblk = 0, w_split = 0, split = 0;
    uint32_t items_T = 0, items_osh = 0;     // the accumulation's work list (bucket_kernels.hpp): entries per lane (0 = not used), population class step
    uint32_t *count = nullptr, *offset = nullptr, *order = nullptr, *bins = nullptr, *sorted = nullptr;
    uint8_t* buckets = nullptr;

    MsmCall(msm_amd_ctx* c, Slot* s, const RunArgs& args) : ctx(c), sl(s), a(args) {}
    int plan();
    int sort();
    int accumulate();
    int reduce();
};

template <class F>
static int submit_msm(msm_amd_ctx* ctx, Slot* sl, const RunArgs& a) {
    MsmCall<F> call(ctx, sl, a);
    sl->deferred_reduce = nullptr;
    if (int rc = call.plan()) return rc == MsmCall<F>::kNothingToQueue ? 0 : rc;
    if (int rc = call.sort()) return rc;
    if (int rc = call.accumulate()) return rc;
    // The reduction has two forms (reduce()): one that hides behind the next call's accumulation, one for a call that nothing follows.
    // Which of them is right is not known here: the first call of a burst looks lone, the last one looks followed (the driver's 20 timed
    // calls: the first reduction, two launches of the lone form, waited 2.4 ms for register room beside the second accumulation and held up
    // the third submission; the last one ran its long hidden form alone, 1.30 ms where the lone form takes 0.58 --
    // profiles/r04_fill_drain_after_buffers.txt).  So a call in a stream of calls -- others in flight now, or lately -- with its
    // accumulation on stream A leaves its reduction to flush_deferred_reduce(): at the next submission, before that call's kernels,
    // or at the wait for its own ticket, whichever comes first.  (`a` is the slot's copy of the arguments: it lives as long as the call.)
    const bool streaming = call.others_pending || ctx->burst_hint > 0;
    if (call.others_pending) ctx->burst_hint = 8;
    else if (ctx->burst_hint > 0) --ctx->burst_hint;
    // (G1 only: a G2 call's reduction is the longer one, and its successor's submission -- which waits for the call before -- comes too late
    // for it: BLS12-381 G2 n = 2^20: 8.06-8.10 against 7.73-7.79 ms per call over 20 calls, the same over 60; BN254 G2 no difference --
    // profiles/r04_defer_reduce_g2_ab.txt.  Option value 2 defers them too.)
    if (ctx->defer_reduce && streaming && call.acc_shared && !call.split_mode && (lanes_per_point<typename MsmCall<F>::FK>() == 1 || ctx->defer_reduce == 2) && &a == &sl->retry_args) {
        auto keep = std::make_shared<MsmCall<F>>(call);
        sl->deferred_reduce = [keep](bool pending) {
            keep->others_pending = pending;
            return keep->reduce();
        };
        return 0;
    }
    return call.reduce();
}

// Queue a slot's deferred reduction, if it has one (ctx->mu held).  `pending`: other calls are in flight at this moment.
static inline int flush_deferred_reduce(msm_amd_ctx* ctx, Slot* sl, bool pending) {
    if (!sl->deferred_reduce) return 0;
    std::function<int(bool)> f = std::move(sl->deferred_reduce);
    sl->deferred_reduce = nullptr;
    (void)hipSetDevice(ctx->device);
    return f(pending);
}

// ---- the call's plan: inputs on the GPU, window layout, schedule, limits, the slot's tables
template <class F>
int MsmCall<F>::plan() {
    sl->trivial = false;
    sl->raw_out = a.raw_out;
    sl->finish = &finish_msm<F>;
    if (a.n == 0 || a.nbits == 0) {
        sl->trivial = true;
        return kNothingToQueue;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // everything up to the sorted index lists runs on stream S -- or S2 when the accumulation will be small: such calls run their
    // accumulation and reduction on their slot's own stream, several of them in flight, and the ~ten short sort kernels of the next
    // call must not queue behind one of those in a shared hardware queue (8-way shard of n = 2^20: 0.77 -> 0.56 ms per call with
    // both measures; calls with a full-size accumulation were 3-8 % slower on S2, hence two streams)
    // the window size: by size and rank count -- except that option "glv" = 2 (tests, experiments) asks for the GLV schedule wherever it
    // applies, and that schedule is built on 16-bit windows
    auto window_bits = [&]() -> uint32_t {
        if (a.prepared) return a.prepared->c;   // the table fixes the window layout
        // 14-bit windows: G1 of both curves from n = 2^15 to 3 x 2^17, BLS12-381 G2 from 49152 to 2^18 (pick_window_bits)
        constexpr bool g1 = KernelField<F>::chained_variant, bls_g2 = std::is_same<F, Fp2<Bls12381Fp>>::value;
        uint32_t c = pick_window_bits(a.n, a.nbits, ctx->force_c, a.nshards, ctx->no_split2 ? 4096u : (g1 ? 98304u : 65537u), ctx->even_windows != 0,
                                      g1 ? 32768u : (bls_g2 ? 49152u : 0u), g1 ? 393216u : (bls_g2 ? (1u << 18) : 0u));
        if constexpr (GlvField<F>::value) {
            if (!ctx->force_c && ctx->glv_mode == 2 && a.n >= 1024 && a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256) c = 16;
            // bases vouched to lie in the subgroup: from n ~ 180 000 the GLV schedule (8 windows of 16 bits over 2n half scalars) is ahead
            // of the 14-bit windows chosen below 2^18 (n = 2^17: 0.94 against 0.86 ms, 2^18: 1.15 against 1.26)
            if (!ctx->force_c && ctx->glv_mode == 1 && (a.flags & MSM_AMD_BASES_IN_SUBGROUP) && a.nshards == 1 && a.n >= 180000u && a.n <= (1u << 18) &&
                a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256)
                c = 16;
        }
        return c;
    };
    others_pending = false;
    for (uint32_t k = 0; k < ctx->nslots; ++k) others_pending |= (&ctx->slots[k] != sl && ctx->slots[k].pending);
    bool small_est;
    {
        const uint32_t c0 = window_bits(), w0 = a.prepared ? 1u : a.nbits / c0 + 1;   // (prepared bases: one bucket set)
        small_est = (uint64_t)((w0 + a.nshards - 1) / a.nshards) << (c0 - 1) < 2u * 1024u * 64u;
    }
    if (small_est && !ctx->stream_s2) {
        // created on first use: the mere existence of a stream of another priority re-partitions the hardware queues and cost calls
        // with a full-size accumulation ~5 % (measured in round 1; at the end of round 3 a context runs its full-size calls at the same 2.59 ms
        // before and after small calls created this stream, profiles/r03_s2_state.txt); of another priority because only such streams never share a queue with S/A/B
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream_s2, hipStreamNonBlocking, 1));
        // (a stream of another priority has a hardware queue of its own: its scratch like the others', engine_ctx.hpp)
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < MSM_SCRATCH_RESERVE * MSM_SCRATCH_SLOTS + MSM_SCRATCH_MARGIN) return msm_alloc_failed(ctx, "scratch of the second sort stream");
        k_scratch_reserve<<<1, 64, 0, ctx->stream_s2>>>(nullptr, 0, 1);
    }
    st = small_est ? ctx->stream_s2 : ctx->stream_s;
    const bool dev_scalars = (a.flags & MSM_AMD_DEVICE_PTRS) != 0;
    const bool dev_bases = a.prepared || dev_scalars || (a.flags & MSM_AMD_DEVICE_BASES) != 0;
    const bool jac = !a.prepared && (a.flags & MSM_AMD_BASES_JACOBIAN) != 0;
    const size_t base_bytes = (size_t)a.n * (jac ? JSZ : ASZ);
    const size_t scalar_bytes = (size_t)a.n * a.scalar_size;

    if (dev_bases) {
        if (((uintptr_t)a.bases & 15) != 0) {
            msm_set_err(ctx, "device bases pointer must be 16-byte aligned");
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

    g = Geom{};
    g.n = a.n;
    g.scalar_size = a.scalar_size;
    g.bit_lo = a.bit_lo;
    g.nbits = a.nbits;
    g.c = window_bits();
    g.W = a.nbits / g.c + 1;
    g.half = 1u << (g.c - 1);
    g.shard = a.shard;
    g.nshards = a.nshards;
    g.ns = g.W;
    g.merged = 0;
    g.stride = 0;
    g.check_wide = 0;
    // Windows of even width.  With nbits / c + 1 windows of c bits the top one holds nbits % c bits: when that is neither 0 nor the full
    // width, its 2^(nbits % c) buckets are 2^(c - 1 - nbits % c) times as dense as all the others (c = 13 over 256 bits: 256 buckets of
    // n / 256 entries each) and their lanes set the duration of the accumulation -- which is why round 2 found no window size between
    // 9 and 15 bits worth using.  Instead: W = ceil((nbits + 1) / c) windows, the first ns of them c bits wide and signed, the rest
    // c - 1 bits wide and unsigned (no carry out of the top one), ns chosen so that they tile the nbits exactly -- both kinds fill their
    // 2^(c-1) buckets evenly (the layout of the prepared tables, Geom in engine_common.hpp).  Kept for nbits % c == 0 is the old
    // layout: its top window is full, and the carry-only window above it stays empty for scalars whose top bit is clear.
    // What is tiled is the EXPECTED width of the scalars, not the declared one: 32-byte scalars are expected below the group order
    // (255 bits for BLS12-381, 254 for BN254 -- a prover's scalars are reduced), and the 1-2 bits above get one more window, the
    // overflow window, which such scalars leave empty (no entries, no buckets touched, skipped by the reduction).  Tiling all 256
    // bits instead leaves the top window of a 254-bit scalar a quarter used -- a quarter of its buckets, four times as dense: BN254 at
    // n = 2^18 spent 0.95 ms in an accumulation of 0.45 ms' worth of additions, and queued over-long buckets for the tail kernels.
    // The declared width IS tiled once recent calls have put entries into the overflow window (wide_hint; scalars that use all 256
    // bits: with c-bit windows they also carry out of the top one into the carry-only window, whose single bucket then holds n / 2
    // entries -- 4.7 instead of 2.8 ms per MSM at n = 2^20; 1 signed + 16 unsigned 15-bit windows have no such window, for 6 % more
    // additions).  The hint decays over 16 calls without such entries being observable (the wide layout has no overflow window).
    const bool wide = (ctx->wide_hint.load() > 0 || a.full_width) && a.nshards == 1 && !ctx->force_seg;
    // Scalars much NARROWER than their 32 bytes (round 4) -- 128-bit challenges, 64-bit witness values: in the layouts above half of
    // them carry out of their top window into ONE bucket of the next (n / 2 entries: 2.3 instead of 1.4 ms per MSM at n = 2^20 for 128-bit
    // scalars), and a width that is not a multiple of c leaves a top window of a few very dense buckets (200 bits: 5.1 ms against the
    // 2.7 of full-width scalars).  k_tile_hist reports the widest scalar of every call; once two calls in a row were narrower than the
    // group order the context's next calls tile THAT width with the same windows of even width, unsigned at the top, so that
    // nothing carries out.  Safe whatever comes: up to c - 1 bits beyond the hint land in the overflow window (exact); a scalar beyond
    // that is reported by the histogram pass (Geom::check_wide) and the call is run again over the declared width when it is
    // collected (finish_ticket) -- one call pays twice; its width joins the last eight, whose maximum is the hint.  Option "narrow" = 0
    // switches the adaptation off.
    uint32_t narrow = 0;
    if (!wide && !a.full_width && ctx->narrow_mode && a.nshards == 1 && !a.prepared && !ctx->force_seg && a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256) {
        narrow = (uint32_t)std::max(0, ctx->narrow_bits.load());
        if (narrow >= OrderBits<F>::value) narrow = 0;
        if (GlvField<F>::value && ctx->glv_mode == 2) narrow = 0;   // (the GLV schedule forced by option: its eight windows over the half scalars, whatever was observed)
    }
    const uint32_t expected = narrow ? std::max(narrow, 8u) : ((a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256 && !wide) ? OrderBits<F>::value : a.nbits);
    sl->top_is_overflow = false;
    sl->narrow_layout = narrow != 0;
    g.check_wide = narrow != 0;
    // Which calls take windows of even width: those whose window size does not divide the width, and every call whose scalars are
    // expected to FILL the width that is tiled (round 4) -- the observed width, the declared width of scalars that are not 32 bytes long
    // or of a bit range, 256-bit scalars after a wide call: with c-bit windows over a width c divides, half of such scalars carry out
    // of the top window into ONE bucket of a carry-only window (n / 2 entries; 16-byte scalars ran 3.05 ms at n = 2^20, 24-byte ones
    // 4.6).  Only 32-byte scalars below the group order keep c-bit windows where c divides 256: their top bit is clear, nothing carries.
    const bool fills = narrow != 0 || expected == g.nbits;
    if (!a.prepared && ctx->even_windows && fills && g.nbits > g.c) {
        // windows of c and c - 1 bits tile `expected` bits exactly, with an unsigned one at the top, only if W (c - 1) <= expected for
        // W = expected / c + 1 -- always true at 255 bits, not for a narrow width (64 bits at c = 16: five 15-bit windows are 75 bits, the
        // top one would hold 4 bits in 16 very dense buckets): take the widest window at or below the size's that does tile it
        uint32_t cc = g.c;
        while (cc > 3 && (expected / cc + 1) * (cc - 1) > expected) --cc;
        g.c = cc;
        g.half = 1u << (g.c - 1);
    }
    // A bit range no wider than the window (the reference's per-window entry point, fnName_chunk: what its external caller fans out): ONE
    // unsigned window over 2^nbits buckets -- the signed form put half the scalars into the single bucket of a carry window
    // (chunk(0, 16) at n = 2^20: 2.7 ms, an 8-bit chunk 6.6).
    const bool one_window = !a.prepared && ctx->even_windows && g.nbits <= g.c && g.nbits >= 2 && g.nbits + 1 <= (uint32_t)MAX_WINDOW_BITS && !ctx->force_c;
    if (one_window) {
        g.c = g.nbits + 1;
        g.half = 1u << (g.c - 1);
        g.W = 1;
        g.ns = 0;
    }
    // (c-bit windows that divide the declared width keep the old layout while its top real window is at least half used by scalars of the
    // expected width -- BLS12-381 at c = 16: 15 of 16 bits, signed: all 2^15 buckets; BN254 at c = 16 has 14 bits there, half the buckets
    // twice as dense, and takes the even layout: 14 windows of 16 bits, 2 of 15)
    const bool top_thin = g.nbits % g.c == 0 && expected < g.nbits && expected % g.c != 0 && expected % g.c < g.c - 1;
    if (one_window) {
        // (laid out above: one unsigned window, no carry, no overflow window)
    } else if (!a.prepared && ctx->even_windows && g.c >= 3 && g.nbits > g.c && (g.nbits % g.c != 0 || wide || top_thin || fills)) {
        g.W = (expected + g.c) / g.c;
        g.ns = expected > g.W * (g.c - 1) ? expected - g.W * (g.c - 1) : 0u;
        if (expected < g.nbits) {   // the overflow window: c - 1 bits from bit `expected` on cover the rest (nbits - expected <= 2 <= c - 1)
            ++g.W;
            sl->top_is_overflow = true;
        }
    } else if (!a.prepared && g.nbits % g.c == 0 && g.nbits > g.c) {
        sl->top_is_overflow = true;   // c-bit windows that divide the width: the top one is the carry-only window
    }
    if (a.prepared) {   // the table's windows: ns signed of c bits, the rest unsigned of c - 1; all into ONE bucket set
        g.W = a.prepared->W;
        g.ns = a.prepared->ns;
        g.merged = 1;
        g.check_wide = 1;
        g.stride = a.prepared->stride;
    }
    // GLV path -- only for callers that vouch for their bases (flag MSM_AMD_BASES_IN_SUBGROUP, or option "glv" = 2): k*P = k1*P + k2*phi(P)
    // with k reduced mod r holds in the order-r subgroup only, while the reference's multiexp is plain integer arithmetic, exact for
    // ANY point of the curve (G1 has a ~2^126 cofactor).  Without the flag the engine keeps the reference's semantics.
    // 2n virtual items with 127-bit signed halves, 8 windows of 16 bits (k_glv_halves).  Same number of bucket
    // additions, half the buckets.  It pays where the bucket reduction dominates (98304 <= n <= 2^18: 10-15 % per call; below, 8-bit
    // windows are faster still); from 2^19 up
    // the accumulation dominates and 8 x 2^15 buckets give only 4096 accumulation waves, which quantise badly over
    // 1024 SIMDs x 3 waves (measured 3-15 % slower), so larger calls keep the 16-window schedule unless option "glv" = 2
    // forces the path (tests, experiments).
    use_glv = false;
    if constexpr (GlvField<F>::value)
        use_glv = !narrow && !a.prepared && (ctx->glv_mode == 2 || (ctx->glv_mode == 1 && (a.flags & MSM_AMD_BASES_IN_SUBGROUP))) && g.c == 16 && a.scalar_size == 32 && a.bit_lo == 0 && a.nbits == 256 && a.n >= 1024 &&
                  (a.n <= (1u << 18) || ctx->glv_mode == 2) && a.n <= (1u << 26) && (((uintptr_t)d_scalars & 15) == 0);
    nv = use_glv ? 2 * a.n : a.n;   // items the sort and the accumulation see
    if (use_glv) {
        g.n = nv;
        g.nbits = 128;
        g.W = 8;
        g.ns = 8;   // eight signed 16-bit windows over the 128-bit halves, whatever layout the 256-bit scalar would have taken
        sl->top_is_overflow = false;   // (127-bit magnitudes: the top window is an ordinary one)
    }
    g.Wl = 0;
    for (uint32_t w = 0; w < g.W; ++w)
        if (w % g.nshards == g.shard) ++g.Wl;
    if (g.merged) g.Wl = 1;
    // entries one owned bucket set can receive: one per item -- one per item AND window when the windows share their buckets
    const uint64_t nve64 = g.merged ? (uint64_t)g.W * nv : nv;
    if (nve64 >= (1ull << 31) || (g.merged && (uint64_t)g.W * g.stride >= (1ull << 31))) {
        msm_set_err(ctx, "windows x points of a prepared table must stay below 2^31");
        return MSM_AMD_EINVAL;
    }
    nve = (uint32_t)nve64;
    // (pair, window) positions -- Mscan, offset, delta, the indices into tmp/sorted -- are 32-bit: the entries of one pass must
    // number fewer than 2^32 (17 windows x 2^27 points fit; 33 windows of 64-byte scalars or a forced narrow window would wrap)
    if ((uint64_t)g.Wl * nve >= (1ull << 32)) {
        msm_set_err(ctx, "windows x points of one pass must stay below 2^32 (fewer points per call, a wider window, or the synchronous entry point with option \"slice\")");
        return MSM_AMD_EINVAL;
    }
    ctx->geometry[0] = g.c;
    ctx->geometry[1] = g.W;
    ctx->geometry[2] = g.half;
    ctx->geometry[3] = g.Wl;
    ctx->geometry[4] = ctx->geometry[5] = 0;
    sl->g = g;
    if (g.Wl == 0) {
        sl->trivial = true;
        return kNothingToQueue;
    }
    NB = g.Wl * g.half;

    HIPCHK(ctx, hipEventRecord(sl->ev[0], st));
    // ---- Jacobian bases -> affine
    if (jac) {
        if (int rc = ensure(ctx, sl->d_affine, (size_t)a.n * ASZ)) return rc;
        uint32_t lanes = (a.n + jacobian_to_affine_points_per_lane<F>() - 1) / jacobian_to_affine_points_per_lane<F>();
        k_jacobian_to_affine<F><<<(lanes + 255) / 256, 256, 0, st>>>(d_bases, a.n, (uint8_t*)sl->d_affine.p);
        d_bases = (const uint8_t*)sl->d_affine.p;
    }
    d_xbeta = nullptr;
    if constexpr (GlvField<F>::value) {
        if (use_glv) {
            if (int rc = ensure(ctx, sl->d_xbeta, (size_t)a.n * sizeof(F))) return rc;
            k_glv_xbeta<F><<<(a.n + 255) / 256, 256, 0, st>>>(d_bases, a.n, (uint8_t*)sl->d_xbeta.p);
            d_xbeta = (const uint8_t*)sl->d_xbeta.p;
        }
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[1], st));

    if (int rc = ensure(ctx, sl->d_count, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_offset, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_order, (size_t)NB * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_bins, 2 * ORDER_KEYS * 4)) return rc;   // key populations, then the per-key cursors
    if (int rc = ensure(ctx, sl->d_sorted, (size_t)g.Wl * nve * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_flags, 16)) return rc;
    if (int rc = ensure(ctx, sl->d_buckets, (size_t)NB * BSZ)) return rc;
    count = (uint32_t*)sl->d_count.p;
    offset = (uint32_t*)sl->d_offset.p;
    order = (uint32_t*)sl->d_order.p;
    bins = (uint32_t*)sl->d_bins.p;
    sorted = (uint32_t*)sl->d_sorted.p;

    nblk = (a.n + 255) / 256;
    oblk = std::max(1u, std::min(512u, NB / 2048u));
    // A lone call (nothing else in flight) is latency-bound at its end: the bucket reduction is ~40 dependent additions deep
    // on one wave per SIMD.  Optionally (option "split") its windows are handled as two groups, the reduction of the first
    // group running under the accumulation of the second.  Measured at n = 2^20 this LOSES (6.3 vs 4.8 ms): the 256-VGPR
    // reduction waves leave room for only one accumulation wave per SIMD, and two launches have two tails.  Kept for
    // experiments, off by default; overlapping across calls (slots) is what pays.  (Round 4, with the plane kernels reducing both
    // groups: still 3.74 against 3.14 ms at n = 2^20, 11.98 against 11.25 at 2^22 -- tools/runs/r04_split_planes.sh.)
    split_mode = !others_pending && !ctx->no_split && g.Wl >= 6 && NB >= 8u * 65536u;
    w_split = split_mode ? g.Wl - std::max(2u, (g.Wl * 3u) / 10u) : g.Wl;   // windows in the first group
    split = w_split * g.half;                                               // first bucket of the second group
    // The work list (k_accumulate_items, bucket_kernels.hpp): lanes per bucket in proportion to its population, every lane at most T
    // entries.  For launches that do not fill the chip several times over anyway -- small n, the windows one rank of a sharded call
    // owns -- where the duration of the single round of waves is its longest lane.  T brings the launch to ~320 k lanes (five waves per
    // SIMD) and is at least 8 (a lane's bucket write and the join's full addition want some additions to amortise over); buckets
    // whose mean needs more than ITEMS_NI_MAX / 2 lanes (a handful of very long buckets: small n in 8-bit windows) keep the sliced
    // launch with its tree join.  Option "items": 0 = by this rule, 1 = wherever the mean allows, 2 = never.
    items_T = items_osh = 0;
    ctx->last_items_T = 0;
    {
        const uint64_t entries = (uint64_t)g.Wl * nve;
        const uint32_t lanes_goal = KernelField<F>::chained_variant ? ctx->splitk_lanes : ctx->splitk_lanes / 3u;
        const uint32_t mean = (nve + g.half - 1) / g.half;
        uint32_t T = (uint32_t)std::max<uint64_t>(8, (entries + lanes_goal - 1) / lanes_goal);
        if (ctx->items_t) T = ctx->items_t;
        const bool fits = (mean + T - 1) / T <= ITEMS_NI_MAX / 2 && T <= 512;
        const bool wanted = ctx->items_mode == 1 || (ctx->items_mode == 0 && NB < lanes_goal);
        if (fits && wanted && !split_mode && !ctx->no_split2 && !ctx->splitk && !ctx->force_seg) {
            items_T = T;
            ctx->last_items_T = T;
            while (((uint64_t)(ITEMS_CLASSES - 1) << items_osh) < (uint64_t)ITEMS_NI_MAX * T) ++items_osh;   // every population the lanes cover lies inside a class
        }
    }
    return 0;
}

// ---- stages 1-3: the bucket sort (sort_kernels.hpp) on the sort stream
template <class F>
int MsmCall<F>::sort() {

    // ---- 1. populations of the (owned window, coarse bin) pairs -- digits recomputed from the scalars, never stored
    SortGeom sg;
    // tiles: 4096 items (1024 lanes x 4) from n = 2^19 -- runs of tile / cb = 32 entries (one 128-byte line) per (tile, window,
    // bin), one reservation per run; smaller inputs take tiles of 1024 (256 lanes x 4) so that more than a handful of CUs work
    // The wide tiles are the faster sort on an idle chip (n = 2^20: 128 against 156 us for the stage), but a 1024-lane workgroup with
    // 100 KB of LDS has to wait for a whole CU to drain of the long-running accumulation waves of the calls already in flight: with
    // another call pending the small tiles give the better THROUGHPUT up to n = 2^22 (n = 2^20: 2.59-2.60 against 2.62-2.63 ms per
    // MSM, 2^22: 10.80 against 10.95; equal from 2^24), and for the ranks of a many-way shard, which keep four calls in flight, at
    // every size (8 ranks: n = 2^20 0.74 -> 0.70 ms per call, 2^22 2.19 -> 2.00, 2^26 24.7 -> 24.2; 2 % the other way at 4 ranks).
    const bool wide = ctx->tile_mode ? ctx->tile_mode == 2 : (nv >= (1u << 19) && g.nshards <= 4 && !(others_pending && g.nshards == 1 && nv < (1u << 23)));
    // fine key: 8 bits up to c = 17 (at c = 16: 128 coarse bins, spans of n / 128 entries -- one workgroup reads a span once up
    // to n = 2^20 -- and runs of tile / 128 entries in the partition), then 8 coarse bits and the rest fine
    sg.fbits = g.c - 1 <= 16u ? std::min(g.c - 1, 8u) : g.c - 1 - 8u;
    // Round 4: what a (window, coarse bin) span may hold and still be read ONCE by one workgroup grew from 9216 to 36864 entries
    // (k_fine_small on 1024 lanes, FINE_CAP_WIDE), and the partition may cut 512 coarse bins (tiles of 8192 scalars, two windows per
    // phase: the runs keep their 16 entries).  Spans that fit are sorted in one pass -- read, LDS permutation, one coalesced stream out --
    // instead of the three passes and 16-byte runs of the chunk-parallel kernels: n = 2^22 at c = 16 (128 bins), 2^23 at c = 19 (256),
    // 2^24 at c = 20 (512), the merged windows of a prepared table at n = 2^20 (512).  Beyond (n >= 2^25) the chunk-parallel kernels
    // stay, on 512 bins where the window allows it: half as many fine buckets per chunk, runs twice as long.  Option "sort_geom" = 1
    // keeps the round-3 geometry (256 bins, 9216-entry spans).
    bool fine_wide = false, tile8k = false;
    {
        auto fits = [&](uint32_t fb, uint32_t cap) {
            const uint64_t m = (uint64_t)nve >> (g.c - 1 - fb);
            return m + 6 * (uint64_t)std::ceil(std::sqrt((double)m)) <= cap && ((size_t)2 * (1u << fb) + cap) * 4 + 8192 <= 160 * 1024;
        };
        const bool merged_ok = !g.merged || g.stride % 8192u == 0;   // (a tile of scalars must not straddle an index slice: rows of the table are `stride` apart)
        // (512 bins double the histogram's counters and their flush: below ~2^24 entries -- a prepared table at n = 2^20 -- the round-3
        // shape stays ahead, 0.22 against 0.235 ms for the stage)
        const bool may8k = wide && g.c - 1 > 16u && g.c <= 20 && !use_glv && merged_ok && ctx->sort_geom != 1 && (nve >= (1u << 24) || ctx->sort_geom >= 2);
        if (ctx->sort_geom >= 2 && may8k) {   // tests: the 512-bin partition (and, = 3, the wide fine sort) at sizes an oracle can follow
            sg.fbits = g.c - 1 - 9u;
            tile8k = true;
            fine_wide = ctx->sort_geom == 3 || (!fits(sg.fbits, FINE_CAP) && fits(sg.fbits, FINE_CAP_WIDE));
        } else if (ctx->sort_geom != 1 && wide && !fits(sg.fbits, FINE_CAP)) {
            if (fits(sg.fbits, FINE_CAP_WIDE)) {
                fine_wide = true;
            } else if (may8k && fits(g.c - 1 - 9u, FINE_CAP_WIDE)) {
                sg.fbits = g.c - 1 - 9u;
                fine_wide = tile8k = true;
            } else if (may8k) {
                sg.fbits = g.c - 1 - 9u;
                tile8k = true;
            }
        }
    }
    sg.cbits = g.c - 1 - sg.fbits;
    sg.cb = 1u << sg.cbits;
    sg.nbins = g.Wl * sg.cb;
    // 4-byte partition entries {fine key | sign | index} at every size: the index field takes what the fine key and the sign leave
    // of 32 bits; wider indices (n > 2^23 at c = 16, the W x n table indices of prepared bases) put their HIGH bits into the bin --
    // S sub-bins per (window, coarse bin), see SortGeom
    const uint64_t max_index = g.merged ? (uint64_t)g.W * g.stride : (uint64_t)nv;
    sg.idxbits = 1;
    while ((1ull << sg.idxbits) < max_index) ++sg.idxbits;
    sg.idxbits = std::min(sg.idxbits, 31u - sg.fbits);
    sg.S = (uint32_t)((max_index + (1ull << sg.idxbits) - 1) >> sg.idxbits);
    if (sg.S == 0) sg.S = 1;
    if (sg.S > 2048u) {
        msm_set_err(ctx, "too many points for one pass of the sort (index slices > 2048): fewer points per call");
        return MSM_AMD_EINVAL;
    }
    const size_t nsub = (size_t)sg.nbins * sg.S;   // bins the partition scatters into
    const uint32_t tile = wide ? 4096u : 1024u;
    const uint32_t ntile = (nv + tile - 1) / tile;
    // chunks of the spans that are cut up (k_big_*): 8192 entries for 1024-lane workgroups on an idle chip, 4096 for 256-lane ones
    // while other calls are in flight (sort_kernels.hpp)
    // (measured, ms per MSM with two calls in flight, small | large chunks: n = 2^22 10.74 | 10.89, prepared 9.14 | 9.26, 2^23 19.32 | 19.36,
    // 2^24 36.9 | 36.5, profiles/r03_chunk_mode.txt)
    const bool small_chunks = ctx->chunk_mode ? ctx->chunk_mode == 1 : (others_pending && a.n < (1u << 24));
    const uint32_t chunk = small_chunks ? (uint32_t)BIG_CHUNK_SMALL : (uint32_t)BIG_CHUNK;
    const size_t max_chunks = ((size_t)g.Wl * nve) / chunk + sg.nbins + 1;
    if (int rc = ensure(ctx, sl->d_tmp, (size_t)g.Wl * nve * 4)) return rc;
    // (a buffer that grows may come back at the same address: "known to be zero" is keyed on the buffer's allocation count as
    // well, ZeroMark in engine_ctx.hpp)
    if (int rc = ensure(ctx, sl->d_binhist, nsub * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_binstart, (nsub + 1) * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_bincur, nsub * 4)) return rc;
    if (int rc = ensure(ctx, sl->d_cpre, (std::max<size_t>(sg.nbins, 64) + 1) * 4)) return rc;   // (also the 64 range sums of the parallel bin scan)
    if (int rc = ensure(ctx, sl->d_desc, max_chunks * 8)) return rc;
    if (int rc = ensure(ctx, sl->d_bcursor, (size_t)NB * 4)) return rc;
    uint32_t* binhist = (uint32_t*)sl->d_binhist.p;
    uint32_t* binstart = (uint32_t*)sl->d_binstart.p;
    uint32_t* bincur = (uint32_t*)sl->d_bincur.p;
    uint32_t* cpre = (uint32_t*)sl->d_cpre.p;
    uint2* desc = (uint2*)sl->d_desc.p;
    uint32_t* bcursor = (uint32_t*)sl->d_bcursor.p;
    // where the digits come from: 0 = any scalar size / bit range (bits read from memory per window), 8 = whole 32-byte scalars in
    // registers, 4 = the 16-byte halves of the GLV schedule
    const uint8_t* d_items = d_scalars;
    int nw = 0;
    if (use_glv) {
        if (int rc = ensure(ctx, sl->d_halves, (size_t)nv * 16)) return rc;
        k_glv_halves<<<nblk, 256, 0, st>>>(d_scalars, a.n, (uint4*)sl->d_halves.p);
        d_items = (const uint8_t*)sl->d_halves.p;
        nw = 4;
    } else if (a.scalar_size == 32 && g.bit_lo == 0 && g.nbits == 256 && (((uintptr_t)d_scalars & 15) == 0)) {
        nw = 8;
    }
    const size_t hist_lds = (size_t)(g.merged ? g.W : g.Wl) * sg.cb * 4;   // one row per window
    if (hist_lds > 64 * 1024) {
        msm_set_err(ctx, "too many windows x coarse bins for the histogram kernel (use a wider window)");
        return MSM_AMD_EINVAL;
    }
    // binhist is all zero between calls: zeroed when (re)allocated, dirtied by k_tile_hist in [0, nbins), zeroed again by k_bin_scan
    if (!sl->binhist_zero.holds(sl->d_binhist)) HIPCHK(ctx, hipMemsetAsync(binhist, 0, sl->d_binhist.cap, st));
    sl->binhist_zero.clear();   // until this submission has got as far as k_bin_scan
    if (!sl->flags_zero.holds(sl->d_flags)) HIPCHK(ctx, hipMemsetAsync(sl->d_flags.p, 0, 16, st));
    sl->flags_zero.clear();   // until this submission has got as far as k_collect_tail
#define MSM_TILE_HIST(NW, BL) k_tile_hist<NW, 4, BL><<<ntile, BL, hist_lds, st>>>(d_items, g, sg, binhist, (uint32_t*)sl->d_flags.p)
    // (one-wave workgroups for the histogram of a call that runs beside another call's accumulation -- a 256-lane workgroup needs a wave slot
    // on all four SIMDs of a CU at once, the accumulation refills slots singly -- were tried: the next accumulation starts 45 us sooner, and
    // the running one loses 230 us to the waves that took its slots: 2.71-2.75 against 2.55-2.58 ms per MSM, profiles/r04_hist_one_wave_tried.txt)
    if (wide) {
        if (nw == 8) MSM_TILE_HIST(8, 1024);
        else if (nw == 4) MSM_TILE_HIST(4, 1024);
        else MSM_TILE_HIST(0, 1024);
    } else {
        if (nw == 8) MSM_TILE_HIST(8, 256);
        else if (nw == 4) MSM_TILE_HIST(4, 256);
        else MSM_TILE_HIST(0, 256);
    }
#undef MSM_TILE_HIST
    HIPCHK(ctx, hipEventRecord(sl->ev[2], st));
    // ---- 2. bin starts
    if (nsub > 32768 && ctx->sort_geom != 1) {   // many bins (large n): the scan by up to 64 workgroups, two short launches (cpre is free until the fine sort plans its chunks)
        const uint32_t sgrid = (uint32_t)std::min<size_t>(64, (nsub + 4095) / 4096);
        k_bin_scan_sums<<<sgrid, 1024, 0, st>>>(binhist, sg.nbins, sg.S, cpre);
        k_bin_scan_ranges<<<sgrid, 1024, 0, st>>>(binhist, sg.nbins, sg.S, cpre, binstart, bincur, bins, 2u * (uint32_t)ORDER_KEYS);
    } else {
        k_bin_scan<<<1, 1024, 0, st>>>(binhist, sg.nbins, sg.S, binstart, bincur, bins, 2u * (uint32_t)ORDER_KEYS);
    }
    sl->binhist_zero.set(sl->d_binhist);
    HIPCHK(ctx, hipEventRecord(sl->ev[3], st));
    // ---- 3. partition by coarse bin, fine sort inside each bin (emits count/offset), ordering by population
    // windows per phase: as many as the staged group fits in LDS (4096-item tiles: four windows of 4-byte entries = 100 KB; the CU has
    // 160 KB and hosts one 1024-lane workgroup)
#define MSM_TILE_SCATTER(NW, BL, E, GW)                                                                                              \
do {                                                                                                                             \
    constexpr size_t lds_ = tile_scatter_lds<E>(4 * BL, GW, BL);                                                                  \
    if (lds_ > 64 * 1024)                                                                                                         \
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_scatter<NW, 4, BL, E, GW>),                         \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_));                                  \
    k_tile_scatter<NW, 4, BL, E, GW><<<ntile, BL, lds_, st>>>(d_items, g, sg, bincur, (E*)sl->d_tmp.p);                           \
} while (0)
#define MSM_TILE_SCATTER_E(NW, BL)          \
do {                                    \
    MSM_TILE_SCATTER(NW, BL, uint32_t, 4);          \
} while (0)
    if (tile8k) {   // 8192 scalars per tile, two windows per phase: 512 coarse bins with runs of 16 entries
        constexpr size_t lds8 = tile_scatter_lds<uint32_t>(8192, 2, 1024);
        const uint32_t ntile8 = (nv + 8191u) / 8192u;
        if (nw == 8) {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_scatter<8, 8, 1024, uint32_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
            k_tile_scatter<8, 8, 1024, uint32_t, 2><<<ntile8, 1024, lds8, st>>>(d_items, g, sg, bincur, (uint32_t*)sl->d_tmp.p);
        } else {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tile_scatter<0, 8, 1024, uint32_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
            k_tile_scatter<0, 8, 1024, uint32_t, 2><<<ntile8, 1024, lds8, st>>>(d_items, g, sg, bincur, (uint32_t*)sl->d_tmp.p);
        }
    } else if (wide) {
        if (nw == 8) MSM_TILE_SCATTER_E(8, 1024);
        else if (nw == 4) MSM_TILE_SCATTER_E(4, 1024);
        else MSM_TILE_SCATTER_E(0, 1024);
    } else {
        if (nw == 8) MSM_TILE_SCATTER_E(8, 256);
        else if (nw == 4) MSM_TILE_SCATTER_E(4, 256);
        else MSM_TILE_SCATTER_E(0, 256);
    }
#undef MSM_TILE_SCATTER_E
#undef MSM_TILE_SCATTER
    const uint32_t nf = 1u << sg.fbits;
    const size_t bnd_lds = sg.S > 1 ? ((size_t)sg.S + 1) * 4 : 0;   // sub-bin boundaries of one span
    const uint32_t cap = fine_wide ? (uint32_t)FINE_CAP_WIDE : (uint32_t)FINE_CAP;   // what one fine-sort workgroup reads in one go
    const size_t small_lds = ((size_t)2 * nf + cap) * 4 + bnd_lds, nf_lds = (size_t)nf * 4, big_lds = ((size_t)2 * nf + 2 * chunk) * 4 + bnd_lds;
    // Spans up to `cap` entries: one workgroup each, read once.  Longer ones -- large n, the dense bins of a partly used top
    // window -- are cut into chunks that workgroups sort in parallel (k_big_*, five more launches); when none is expected those
    // launches are skipped and an unexpected long span (adversarial scalars) is sorted by its one workgroup, chunk by chunk.
    // (a prepared table's windows all fill the bucket set evenly: no partly used top window)
    const uint32_t tb = (g.merged || g.ns != g.W) ? 0u : (g.nbits > (g.W - 1) * g.c ? g.nbits - (g.W - 1) * g.c : 0u);   // scalar bits in the top window
    const uint64_t mean_span = (uint64_t)nve / sg.cb;
    const bool expect_long = mean_span + 6 * (uint64_t)std::ceil(std::sqrt((double)mean_span)) > (uint64_t)cap || (tb > 0 && tb < g.c - 1 && (uint64_t)nv > (uint64_t)cap) || ctx->sort_big == 1 || (ctx->sort_big == 0 && ctx->skew_hint.load() > 0);
    auto fine_small = [&](bool long_here) -> int {
        if (fine_wide) {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine_small<uint32_t, true, FINE_BLOCK_WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine_small<uint32_t, false, FINE_BLOCK_WIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
            if (long_here) k_fine_small<uint32_t, true, FINE_BLOCK_WIDE><<<sg.nbins, FINE_BLOCK_WIDE, small_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, sg, g.half, sorted, count, offset);
            else k_fine_small<uint32_t, false, FINE_BLOCK_WIDE><<<sg.nbins, FINE_BLOCK_WIDE, small_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, sg, g.half, sorted, count, offset);
            return 0;
        }
        if (small_lds > 64 * 1024) {   // beyond the default dynamic-LDS limit (the CU has 160 KB)
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine_small<uint32_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine_small<uint32_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
        }
        if (long_here) k_fine_small<uint32_t, true><<<sg.nbins, SORT_BLOCK, small_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, sg, g.half, sorted, count, offset);
        else k_fine_small<uint32_t, false><<<sg.nbins, SORT_BLOCK, small_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, sg, g.half, sorted, count, offset);
        return 0;
    };
    if (!expect_long || ctx->sort_big == 2) {
        if (int rc = fine_small(true)) return rc;
    } else {
        if (big_lds > 64 * 1024) {   // beyond the default dynamic-LDS limit (the CU has 160 KB)
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_scatter<uint32_t, 1024, BIG_CHUNK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big_scatter<uint32_t, 256, BIG_CHUNK_SMALL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
        }
        const uint32_t big_grid = (uint32_t)std::min<size_t>((max_chunks + 7) / 8 * 8, 4096);   // a multiple of 8: one share per XCD
        HIPCHK(ctx, hipMemsetAsync(count, 0, (size_t)NB * 4, st));   // the buckets of cut-up spans are counted with atomics
        k_big_plan<<<1, 1024, 0, st>>>(binstart, sg.nbins, sg.S, chunk, cap, cpre);
        k_big_desc<<<(uint32_t)std::min<size_t>((max_chunks + 255) / 256, 1024), 256, 0, st>>>(binstart, cpre, sg.nbins, sg.S, chunk, desc);
        if (int rc = fine_small(false)) return rc;
        k_big_hist<uint32_t><<<big_grid, SORT_BLOCK, nf_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, cpre, desc, sg, g.half, chunk, count);
        k_big_offsets<<<sg.nbins, SORT_BLOCK, nf_lds, st>>>(binstart, sg, g.half, cap, count, offset, bcursor);
        // one resident wave of workgroups, one per CU, each XCD walking its own eighth of the chunk list in order (see the kernel)
        const uint32_t bigs_grid = (uint32_t)std::min<size_t>((max_chunks + 7) / 8 * 8, 256);
        if (small_chunks) {
            // (256-lane workgroups: several per CU, so the grid is the number of chunks, not one resident wave per CU)
            const uint32_t grid_s = (uint32_t)std::min<size_t>((max_chunks + 7) / 8 * 8, 2048);
            k_big_scatter<uint32_t, 256, BIG_CHUNK_SMALL><<<grid_s, 256, big_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, cpre, desc, sg, g.half, bcursor, sorted);
        } else {
            k_big_scatter<uint32_t, 1024, BIG_CHUNK><<<bigs_grid, 1024, big_lds, st>>>((const uint32_t*)sl->d_tmp.p, binstart, cpre, desc, sg, g.half, bcursor, sorted);
        }
    }
    k_order_hist<<<oblk, 1024, 0, st>>>(count, NB, split, items_osh, bins);
    k_order_scatter<<<oblk, 1024, 0, st>>>(count, NB, split, items_osh, bins, bins + ORDER_KEYS, order);
    if (items_T) {
        if (int rc = ensure(ctx, sl->d_itab, (2 * (size_t)ITEMS_CLASSES + 2) * 4)) return rc;
        k_items_plan<<<1, 1024, 0, st>>>(bins, items_T, items_osh, (uint32_t*)sl->d_itab.p);
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[4], st));
    return 0;
}

template <class F>
int MsmCall<F>::accumulate() {
    // ---- 4. accumulate (stream A, after this call's sort and after the previous call's accumulation).
    // A SMALL accumulation (buckets for fewer than five waves per SIMD: few owned windows of a sharded call, small n) goes to the
    // slot's own stream instead and may overlap the accumulations of the other calls in flight: measured better or equal whenever
    // the launch is short of ~1.5 rounds, also when its buckets are cut into slices (n = 2^20 over 4 ranks: 1.12 vs 1.25 ms per
    // call; n = 2^26 over 8 ranks: 25.2 vs 26.7).
    const uint32_t mean = (nve + g.half - 1) / g.half;
    // lanes per bucket (k_accumulate_slices; 1 = the plain kernel).  Two reasons to cut buckets into slices, both measured on the
    // windows one rank of a sharded call owns (profiles/r02_slices_sweep.txt):
    //  (a) too few buckets to occupy the chip: bring the launch to ~320k lanes (five waves per SIMD, i.e. 1.67 rounds of the three
    //      that fit) -- n = 2^20 over 4 ranks, 163840 buckets of 32: 1.12 -> 1.01 ms per call with two lanes per bucket;
    //  (b) long buckets: chains of ~128 entries instead of thousands, as long as the launch stays below 2^21 lanes -- n = 2^26 over
    //      8 ranks, 65536 buckets of 2048: 44.7 ms with one lane per bucket, 26.4 with 3 (one round of three waves per SIMD),
    //      25.0 with 8, 24.2 with 16.
    // Every lane keeps at least 8 entries (the join costs one full addition per slice).  Option "splitk" forces a value,
    // "split2" = 0 keeps every bucket on one lane.
    auto lanes_per_bucket = [&](uint32_t nbk) -> uint32_t {
        if (ctx->no_split2 || nbk == 0) return 1u;
        // G2 (a point on a lane pair, two waves per SIMD): a third of the buckets G1 needs to fill the chip
        const uint32_t lanes_goal = KernelField<F>::chained_variant ? ctx->splitk_lanes : ctx->splitk_lanes / 3u;
        const uint32_t lanes_max = KernelField<F>::chained_variant ? (1u << 21) : (1u << 21) / 3u;
        uint32_t k = ctx->splitk;
        if (!k) {
            // launches that occupy the chip anyway gain nothing (bn128 n = 2^24, 524288 buckets of 512: 21.9 ms/step with one lane per
            // bucket, 22.7 with 4; bls12381 n = 2^24 c = 19: 40.2 vs 44.3)
            if (nbk >= lanes_goal) return 1u;
            const uint32_t k_occupancy = lanes_goal / nbk;
            const uint32_t k_chain = std::min<uint32_t>(mean / 128u, lanes_max / nbk);
            k = std::min<uint32_t>(std::max(k_occupancy, k_chain), MAX_SLICES);
        }
        while (k > 1 && mean / k < 8) --k;
        if (k >= 8 && !ctx->splitk)   // the tree join (k_join_slices_tree) takes powers of two
            while (k & (k - 1)) k &= k - 1;
        return std::max(k, 1u);
    };
    // segments are expected for the tail kernels when recent calls queued some (skewed scalars), when the segment length is forced,
    // and when the top window is only partly filled (few, dense buckets; the last rank of a 254-bit field)
    const bool even_layout = g.merged || g.ns != g.W;   // every window fills its buckets evenly: no partly used top window
    const bool tail_expected = ctx->skew_hint.load() > 0 || ctx->force_seg != 0 || (!even_layout && (g.nbits % g.c) != 0);
    const uint32_t k_first = lanes_per_bucket(split), k_second = split < NB ? lanes_per_bucket(NB - split) : 1u;
    const bool small_acc = ctx->acc_stream ? ctx->acc_stream == 1 : (NB < 5u * 1024u * 64u && !split_mode);
    st = small_acc ? sl->stream_b : ctx->stream;   // (the slot's B stream: its reduction follows on the same stream anyway)
    acc_shared = !small_acc;
    HIPCHK(ctx, hipStreamWaitEvent(st, sl->ev[4], 0));
    HIPCHK(ctx, hipEventRecord(sl->ev[9], st));
    // segment length: mean bucket population + 8 standard deviations of a Poisson bucket, at least 32: random inputs
    // practically never exceed it (a handful of segments queued for the tail kernels costs a serial chain of `seg` additions
    // on one lane, ~0.6 ms at seg = 32, which a 2x-the-mean rule paid at mean = 16).  Longer buckets -- adversarial scalars, or
    // the few buckets of a narrow top window when the scalar width is not a multiple of c -- are cut into segments, which
    // bounds the longest dependent chain of additions.
    // A top window that the scalar bits only partly fill (c = 20 over 256 bits: 16 bits of 20) uses 2^tb of its buckets, each
    // 2^(c-1-tb) times denser than the rest.  When those are many (>= 4096: 64 waves) they run as ordinary lanes with longer
    // chains -- started first, since buckets are ordered by falling population -- rather than as segments for the tail kernels
    // (one workgroup iteration per over-long bucket: 32768 of them cost 40 ms at n = 2^24, profiles/r02_sweep_c_first.txt).
    uint32_t seg = ctx->force_seg ? ctx->force_seg : std::max(32u, mean + 8u * (uint32_t)std::ceil(std::sqrt((double)mean)) + 8u);
    if (!ctx->force_seg) {
        // Buckets that are denser than the mean without being pathological: (a) scalars one bit shorter than their declared width
        // (a 254-bit group order in 32 bytes) fill the top window's buckets twice as densely; (b) a top window that the scalar bits
        // only partly fill (c = 20 over 256 bits: 16 of 20) uses 2^tb of its buckets, 2^(c-1-tb) times denser -- twice that when the
        // top bit is clear.  Such buckets are many, equally long, and start first (population order): they are lanes like any other,
        // as long as their chain stays below the launch's own duration (cap: 80 % of a lane slot's share of the additions).
        auto bound = [](uint64_t d) { return (uint32_t)std::min<uint64_t>(d + 8u * (uint64_t)std::ceil(std::sqrt((double)d)) + 8u, 1u << 24); };
        // Case (a) is not capped (round 3): the cap is about how long a chain may get, and twice the mean is where such buckets END --
        // capping the segment length at 88 where they hold 64 +- 8 entries sent a few dozen of them per call through the tail kernels
        // (BN254 G2 n = 2^20 with scalars below its 254-bit order: 4.41 ms per MSM against 3.87 with 255-bit scalars).
        uint32_t seg_hi = 0;
        const uint32_t tb = even_layout ? 0u : (g.nbits > (g.W - 1) * g.c ? g.nbits - (g.W - 1) * g.c : 0u);   // scalar bits in the top window
        if (tb >= 12 && tb < g.c - 1) seg_hi = bound(2ull * (((uint64_t)nv + (1ull << tb) - 1) >> tb));
        const uint32_t cap = (uint32_t)std::min<uint64_t>(((uint64_t)g.Wl * nve) / 245760u, 1u << 20);
        seg = std::max(std::max(seg, bound(2ull * mean)), std::min(seg_hi, cap));
    }
    const uint32_t seg_min = (seg + std::max(k_first, k_second) - 1) / std::max(k_first, k_second);
    // Hot buckets (engine_common.hpp: more than HOT_FACTOR x seg entries -- a witness full of ones puts 0.3 n entries into ONE bucket) are
    // cut into segments of seg_t entries instead of seg: as short as the segment lists allow (2^20 segments at most: 16 entries up to n = 2^19,
    // 17 at n = 2^20, `seg` itself from n ~ 2^23 on, where the ordinary lanes run that long anyway).
    const uint64_t all_entries = (uint64_t)g.Wl * nve;
    const uint32_t seg_t = ctx->no_hot ? seg : (uint32_t)std::min<uint64_t>(seg, std::max<uint64_t>(16, (all_entries + (1u << 20) - 1) >> 20));
    const size_t max_segs = (size_t)(all_entries / seg_min + all_entries / seg_t + 2);
    if (int rc = ensure(ctx, sl->d_tailctl, HOT_PART_OFFSET + (size_t)HOT_MAX * HOT_GROUPS * BSZ)) return rc;   // counters, hot list, hot partial sums
    if (int rc = ensure(ctx, sl->d_work, max_segs * 16)) return rc;
    if (int rc = ensure(ctx, sl->d_biglist, max_segs * 16)) return rc;
    if (int rc = ensure(ctx, sl->d_segout, max_segs * BSZ)) return rc;
    // the work list: at most entries / T + buckets lanes.  Only the lanes of multi-lane buckets leave their sums in `slices`, and they
    // come first: such a bucket holds more than T - 2^osh >= T / 2 entries (its population class reaches beyond T), so there are fewer than
    // 2 x entries / T of them, with fewer than 3 x entries / T lanes
    const uint32_t max_items = items_T ? (uint32_t)std::min<uint64_t>(all_entries / items_T + NB, 0xffffffc0ull) : 0u;
    const uint32_t max_multi = items_T ? (uint32_t)std::min<uint64_t>(2 * (all_entries / items_T) + 1, NB) : 0u;
    if (items_T) {
        if (int rc = ensure(ctx, sl->d_slices, (size_t)std::min<uint64_t>(max_items, 3 * (all_entries / items_T) + 64) * BSZ)) return rc;
    } else if (k_first > 1 || k_second > 1)
        if (int rc = ensure(ctx, sl->d_slices, std::max((size_t)split * k_first, (size_t)(NB - split) * k_second) * BSZ)) return rc;
    uint32_t tail_block = 256;
    while ((size_t)tail_block * BSZ > 64 * 1024 && tail_block > 64) tail_block /= 2;
    buckets = (uint8_t*)sl->d_buckets.p;
    // G1 has two builds of the accumulation kernels (bucket_kernels.hpp): the chained one when the launch puts at least two waves
    // on every SIMD, the two-chain one otherwise
    auto accumulate = [&](const uint32_t* ord, uint32_t nbk, uint32_t k) -> int {
        if (!sl->ctl_zero.holds(sl->d_tailctl)) HIPCHK(ctx, hipMemsetAsync(sl->d_tailctl.p, 0, 16, st));
        sl->ctl_zero.clear();   // (a second group of a split call zeroes them itself; k_collect_tail sets the mark again)
        if (items_T) {   // (one group only: plan() keeps the work list off when the call is split)
            const uint32_t* itab = (const uint32_t*)sl->d_itab.p;
            uint8_t* items_out = (uint8_t*)sl->d_slices.p;
            if constexpr (KernelField<F>::chained_variant) {
                const bool chained = ctx->acc_variant ? ctx->acc_variant == 2 : all_entries / items_T >= 2u * 1024u * 64u;
                if (chained) launch_accumulate_items<FK, 1>(st, d_bases, d_xbeta, sorted, offset, count, ord, itab, items_T, items_osh, max_items, buckets, items_out);
                else launch_accumulate_items<FK, 0>(st, d_bases, d_xbeta, sorted, offset, count, ord, itab, items_T, items_osh, max_items, buckets, items_out);
            } else {
                launch_accumulate_items<FK, 0>(st, d_bases, d_xbeta, sorted, offset, count, ord, itab, items_T, items_osh, max_items, buckets, items_out);
            }
            launch_join_items<FK>(st, count, ord, itab, items_T, items_osh, max_multi, items_out, buckets, seg, seg_t, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
            launch_accumulate_tail<FK>(st, d_bases, d_xbeta, sorted, offset, count, buckets, seg, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p,
                                       (uint8_t*)sl->d_segout.p, tail_block, tail_expected);
            return 0;
        }
        const uint32_t seg_lane = (seg + k - 1) / k;
        if constexpr (KernelField<F>::chained_variant) {
            const bool chained = ctx->acc_variant ? ctx->acc_variant == 2 : (uint64_t)nbk * k >= 2u * 1024u * 64u;
            if (k > 1) {
                uint8_t* slices = (uint8_t*)sl->d_slices.p;
                if (chained) launch_accumulate_slices<FK, 1>(st, k, d_bases, d_xbeta, sorted, offset, count, ord, nbk, slices, seg_lane);
                else launch_accumulate_slices<FK, 0>(st, k, d_bases, d_xbeta, sorted, offset, count, ord, nbk, slices, seg_lane);
                launch_join_slices<FK>(st, k, count, ord, nbk, slices, buckets, seg_lane, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
            } else if (chained)
                launch_accumulate<FK, 1>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, seg_t, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
            else
                launch_accumulate<FK, 0>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, seg_t, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
        } else if (k > 1) {
            uint8_t* slices = (uint8_t*)sl->d_slices.p;
            launch_accumulate_slices<FK, 0>(st, k, d_bases, d_xbeta, sorted, offset, count, ord, nbk, slices, seg_lane);
            launch_join_slices<FK>(st, k, count, ord, nbk, slices, buckets, seg_lane, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
        } else {
            launch_accumulate<FK, 0>(st, d_bases, d_xbeta, sorted, offset, count, ord, nbk, buckets, seg, seg_t, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p);
        }
        launch_accumulate_tail<FK>(st, d_bases, d_xbeta, sorted, offset, count, buckets, seg_lane, sl->d_tailctl.p, sl->d_work.p, sl->d_biglist.p,
                                   (uint8_t*)sl->d_segout.p, tail_block, tail_expected);
        return 0;
    };
    if (int rc = accumulate(order, split, k_first)) return rc;
    HIPCHK(ctx, hipEventRecord(sl->ev[8], st));   // first group accumulated
    if (split < NB)
        if (int rc = accumulate(order + split, NB - split, k_second)) return rc;
    HIPCHK(ctx, hipEventRecord(sl->ev[5], st));
    return 0;
}

template <class F>
int MsmCall<F>::reduce() {
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
    struct Shape {
        Plan p0, p1;
        bool planes;
        uint32_t planes_L, planes_mlog;
    };
    // (the shape of the reduction is a function of whether other calls are in flight; a slot sizes its buffers for both forms below)
    auto shape_for = [&](bool pending) -> Shape {
        auto plan = [&](uint32_t nbuckets) {
            Plan p;
            // (wide windows have 16-32 times the buckets: with 2^19 per window even a lone call fills the chip many times over at
            // L = 8, and what counts is again the instruction volume -- n = 2^24, c = 20: reduction 8.6 ms at L = 8, 4.6 at L = 128,
            // profiles/r02_reduce_l_wide_windows.txt -- hence the same rule with a cap of 128 instead of a fixed 8)
            const uint32_t wide_cap = g.c > 16 ? 128u : (pending ? 32u : 8u);
            p.L = pending ? std::min(wide_cap, std::max(1u, (uint32_t)(((uint64_t)g.Wl * g.half) / 16384u)))
                                 : std::min(wide_cap * lanes_per_point<FK>(), std::max(1u, (uint32_t)(((uint64_t)nbuckets * lanes_per_point<FK>() + 65535u) / 65536u)));
            // a power of two (the nearer one): a lane multiplies its run by its first index t*L with double-and-add, and the zero low
            // bits of that index cost no additions (sharded calls: L = 10 -> 8 took 1.25 -> 1.17 ms per call at 4 ranks, 18 -> 16 1.62 -> 1.56 at 2)
            {
                uint32_t f = 1;
                while (f * 2 <= p.L) f *= 2;
                p.L = (2 * p.L >= 3 * f) ? 2 * f : f;
            }
            if (ctx->force_L) p.L = ctx->force_L;   // (a power of two: msm_amd_set_option rejects anything else -- the LDS trees halve)
            p.lanes = (g.half + p.L - 1) / p.L;
            p.rblock = std::min(256u, std::max(64u, p.lanes));
            while ((size_t)p.rblock * BSZ > 64 * 1024 && p.rblock > 64) p.rblock /= 2;   // LDS budget (64 KB default limit)
            p.wgs = (p.lanes + p.rblock - 1) / p.rblock;
            return p;
        };
        Plan p0 = plan(split), p1 = plan(NB - split);
        // Round 4: the reduction as plain sums (k_bucket_planes: no double-and-add on the GPU, the host's Horner pass applies the weights;
        // one wave is the unit of work).  Buckets per lane: a lone call wants one wave per SIMD and the shortest chain (2L + 7 additions, then
        // the window's 6); small calls in flight the same, their slots wait for this chain (n = 2^16, three in flight, L = 2 | 4 | 8 | 16:
        // 0.463 | 0.415 | 0.438 | 0.501 ms per call; k_bucket_reduce 0.484).  The large pipelined calls keep k_bucket_reduce: their reduction waits
        // for register room beside the next call's accumulation either way, and two launches wait twice (G1, n = 2^20: 2.65 against 2.55 ms for
        // BLS12-381, 1.42 against 1.35 for BN254; from n = 2^22 on, and for G2, the plane kernels win again: 10.6 against 10.8 ms, 2.56 against 2.68).
        uint32_t planes_mlog = 0, planes_L = 0;
        // (... as long as that accumulation is the long one: with fewer than 24 entries per bucket on average -- the last call of the same shape
        // says -- it is not, and the reduction's chain is what a slot waits for: witness-like scalars at n = 2^20, profiles/r04_witness_like_sparse.txt)
        const bool sparse = ctx->fill_nb.load() == NB && ctx->fill_mean16.load() < 24u * 16u;
        // (windows above 16 bits, other calls in flight: k_bucket_reduce as well -- n = 2^23, c = 19: 19.3 against 21.5 ms per call; c = 20 even)
        const bool planes_hidden = pending && !sparse &&
                                   (g.c > 16 || ((uint64_t)g.Wl * g.half >= (1u << 19) && a.n < (1u << 21) && lanes_per_point<FK>() == 1));
        bool planes = ctx->reduce_planes && (ctx->reduce_planes == 2 || !planes_hidden) && g.half >= 256;
        if (planes) {
            const uint32_t kl = planes_lane_bits<FK>(), lpw = 1u << kl;
            uint32_t L = std::max(4u, pending && g.c <= 16 ? (sparse ? std::min(8u, p0.L / 2) : p0.L / 2) : p0.L);
            if (ctx->force_L) L = std::max(2u, ctx->force_L);
            while (L > 2 && g.half / L < lpw) L /= 2;
            // the window's second stage is one wave per value, every lane adding up at most four of the window's waves first
            const uint32_t L_rule = L;
            while (g.half / (L * lpw) > 4 * lpw) L *= 2;
            // (a single very wide window -- the merged windows of a prepared table at n = 2^20 -- would need running sums four times as long as
            // k_bucket_reduce's rule gives it: that kernel's chain is the shorter one there, latency 3.08 against 3.68 ms)
            if (L > 2 * L_rule && ctx->reduce_planes != 2) planes = false;
            if (L > 256) planes = false;
            planes_L = L;
            for (planes_mlog = 0; ((uint64_t)L * lpw << planes_mlog) < g.half; ++planes_mlog) {}
            if (planes) {
                p0.L = L;
                p0.wgs = 1 + kl + planes_mlog;   // partials per window
                p1 = p0;                         // (a split call: both window groups the same way)
            }
        }
        return Shape{p0, p1, planes, planes_L, planes_mlog};
    };
    const Shape shape = shape_for(others_pending), other_shape = shape_for(!others_pending);
    Plan p0 = shape.p0, p1 = shape.p1;
    const bool planes = shape.planes;
    const uint32_t planes_L = shape.planes_L, planes_mlog = shape.planes_mlog;
    const uint32_t nparts0 = w_split * p0.wgs, nparts1 = (g.Wl - w_split) * p1.wgs, nparts = nparts0 + nparts1;
    sl->part_off.assign(g.Wl, 0);
    sl->part_cnt.assign(g.Wl, 0);
    for (uint32_t wl = 0; wl < g.Wl; ++wl) {
        sl->part_off[wl] = wl < w_split ? wl * p0.wgs : nparts0 + (wl - w_split) * p1.wgs;
        sl->part_cnt[wl] = wl < w_split ? p0.wgs : p1.wgs;
    }
    sl->plane_shift = -1;
    if (planes)
        for (sl->plane_shift = 0; (1u << sl->plane_shift) < planes_L; ++sl->plane_shift) {}
    // The buffers take the larger of the two forms' needs, whichever form this call runs: a slot that had only seen calls with others in
    // flight used to free and allocate again on its first lone call (three times the partials, the plane sums) -- and a free waits for
    // the device, i.e. for this very call's accumulation: the first of the driver's 20 timed calls after 5 warmup calls lands on such a slot,
    // and its successor was submitted 3 ms late (profiles/r04_fill_drain_before.txt).
    auto planes_bytes = [&](const Shape& sh) { return sh.planes && sh.planes_mlog ? ((size_t)g.Wl << sh.planes_mlog) * (planes_lane_bits<FK>() + 2) * BSZ : (size_t)0; };
    auto parts_of = [&](const Shape& sh) { return (size_t)g.Wl * std::max(sh.p0.wgs, sh.p1.wgs); };
    // (the other form is evaluated on this call's window groups, and a lone call may split them differently: twice its figure covers that;
    // these buffers are small -- n = 2^20: 78 KB of plane sums, 22 KB of partials)
    const size_t planes_need = std::max(planes_bytes(shape), 2 * planes_bytes(other_shape));
    const size_t parts_need = std::max((size_t)nparts, 2 * parts_of(other_shape));
    if (planes_need)
        if (int rc = ensure(ctx, sl->d_planes, planes_need)) return rc;
    if (int rc = ensure(ctx, sl->d_partials, parts_need * XSZ + 64)) return rc;   // the partials, then the eight tail words (k_collect_tail)
    if (sl->h_partials_cap < parts_need * XSZ + 64) {
        if (sl->h_partials) HIPCHK(ctx, hipHostFree(sl->h_partials));
        sl->h_partials = nullptr;
        sl->h_partials_cap = 0;
        HIPCHK(ctx, hipHostMalloc(&sl->h_partials, parts_need * XSZ * 2 + 64, hipHostMallocDefault));
        sl->h_partials_cap = parts_need * XSZ * 2 + 64;
    }
    uint8_t* partials = ctx->zero_copy ? (uint8_t*)sl->h_partials : (uint8_t*)sl->d_partials.p;
    HIPCHK(ctx, hipStreamWaitEvent(sb, sl->ev[8], 0));
    // G1, option "reduce_pair" = 1: the reduction runs every addition on a lane pair (8 product rounds instead of 14, 178 instead of
    // 256 registers).  Measured and left off (profiles/r02_reduce_lane_pair_tried.txt): the same time one call at a time, 4-5 % slower
    // when calls are pipelined (2 x 4500 instead of 7000 instructions per addition)
    auto reduce = [&](hipStream_t s_, uint32_t np, const Plan& pl, const uint8_t* bk, uint8_t* part, uint32_t gb0_) {
        if constexpr (KernelField<F>::chained_variant) {
            if (ctx->reduce_pair) {
                launch_bucket_reduce_pair<FK>(s_, np, pl.rblock, bk, g.half, pl.L, pl.lanes, pl.wgs, part, offset, count, gb0_, NB);
                return;
            }
        }
        launch_bucket_reduce<FK>(s_, np, pl.rblock, bk, g.half, pl.L, pl.lanes, pl.wgs, part, offset, count, gb0_, NB);
    };
    if (planes) launch_bucket_planes<FK>(sb, w_split, g.half, planes_L, planes_mlog, buckets, (uint8_t*)sl->d_planes.p, partials, offset, count, 0, NB);
    else reduce(sb, nparts0, p0, buckets, partials, 0);
    if (nparts1) {
        hipStream_t sc = sl->stream_c;
        HIPCHK(ctx, hipStreamWaitEvent(sc, sl->ev[5], 0));
        if (planes)
            launch_bucket_planes<FK>(sc, g.Wl - w_split, g.half, planes_L, planes_mlog, buckets + (size_t)split * BSZ,
                                     (uint8_t*)sl->d_planes.p + ((size_t)w_split << planes_mlog) * (planes_lane_bits<FK>() + 2) * BSZ, partials + (size_t)nparts0 * XSZ,
                                     offset, count, split, NB);
        else reduce(sc, nparts1, p1, buckets + (size_t)split * BSZ, partials + (size_t)nparts0 * XSZ, split);
        HIPCHK(ctx, hipEventRecord(sl->ev_c, sc));
        HIPCHK(ctx, hipStreamWaitEvent(sb, sl->ev_c, 0));
    }
    HIPCHK(ctx, hipEventRecord(sl->ev[6], sb));
    // one copy: the partials and, behind them, the tail words -- total number of scattered (pair, window) entries = last offset + last
    // count (roofline accounting); the tail counters of the (last) accumulation: buckets that overflowed their segment tell the next
    // calls that the input is skewed; scalars wider than a prepared table's windows; where the top owned window's entries start
    k_collect_tail<<<1, 64, 0, sb>>>(offset, count, NB, (g.Wl - 1) * g.half, (uint32_t*)sl->d_tailctl.p, (uint32_t*)sl->d_flags.p,
                                     reinterpret_cast<uint32_t*>(partials + (size_t)nparts * XSZ));
    sl->ctl_zero.set(sl->d_tailctl);
    sl->flags_zero.set(sl->d_flags);
    if (!ctx->zero_copy) HIPCHK(ctx, hipMemcpyAsync(sl->h_partials, sl->d_partials.p, (size_t)nparts * XSZ + 48, hipMemcpyDeviceToHost, sb));
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
    // Runs WITHOUT the context's lock (round 4: the slot is this thread's while `collecting` is set; the host combine of one call no
    // longer keeps other threads from submitting or collecting theirs -- the libuv workers behind concurrent JS promises).  What the
    // context publishes about its last call -- timings, geometry -- is gathered in locals and copied under the lock at the end.
    float tm[8] = {};
    const Geom& g = sl->g;
    uint32_t geo[6] = {g.c, g.W, g.half, g.Wl, 0, 0};
    auto publish = [&] {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (int i = 0; i < 8; ++i) ctx->timings[i] = tm[i];
        for (int i = 0; i < 6; ++i) ctx->geometry[i] = geo[i];
    };
    if (sl->trivial) {
        host_store_jacobian_normalised(XYZZ<typename HostOf<F>::type>::inf(), out);
        publish();
        return 0;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(sl->ev[7]));
    const uint32_t nparts = sl->nparts;
    uint32_t* h_tail = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(sl->h_partials) + (size_t)nparts * XSZ);
    {
        uint64_t entries = (uint64_t)h_tail[0] + h_tail[1];
        geo[4] = (uint32_t)entries;
        geo[5] = (uint32_t)(entries >> 32);
        // Skewed input (buckets beyond the segment length: scalars wider than the group order, a hot bucket, adversarial data) usually
        // means long spans in the fine sort as well, which the sort only cuts into parallel chunks when it expects them: the next
        // calls of this context do (n = 2^20, 256-bit scalars: scatter stage 1.7 -> 0.3 ms from the second call on).
        if (getenv("MSM_AMD_DEBUG_TAIL")) fprintf(stderr, "msm_amd tail: %u segments, %u big buckets, %u hot\n", h_tail[2], h_tail[3], h_tail[4]);
        if (g.nshards == 1 && sl->NB) {
            ctx->fill_nb.store(sl->NB);
            ctx->fill_mean16.store((uint32_t)std::min<uint64_t>(entries * 16u / sl->NB, 0xffffffffu));
        }
        if (h_tail[3] /* TailCtl::nbig */ > 0) ctx->skew_hint.store(16);
        else if (ctx->skew_hint.load() > 0) ctx->skew_hint.fetch_sub(1);
        // entries in the overflow window (plan()): the scalars are wider than the group order -- the next calls tile the declared width
        if (g.nshards == 1 && !g.merged) {
            const bool over = sl->top_is_overflow && entries > (uint64_t)h_tail[7];   // entries in the overflow window
            // the widest scalar of this call (k_tile_hist; 32-byte scalars read from registers only) joins the last eight: their maximum is
            // what the next calls' windows tile.  A stream that alternates widths therefore keeps the wide layout (no second runs), a
            // stream that turns wider pays ONE second run and follows from the next call on, one that turns narrower follows eight calls later.
            const uint32_t ob = h_tail[8];
            if (ob && ctx->narrow_mode && g.scalar_size == 32 && g.bit_lo == 0) {
                std::lock_guard<std::mutex> hl(ctx->hint_mu);
                ctx->width_hist[ctx->width_count++ % 8u] = (uint16_t)ob;
                uint32_t m = 0;
                for (uint32_t i = 0; i < std::min(ctx->width_count, 8u); ++i) m = std::max<uint32_t>(m, ctx->width_hist[i]);
                ctx->narrow_bits.store(ctx->width_count >= 2 ? (int)m : 0);
            }
            if (sl->narrow_layout) {
                if (h_tail[6]) {   // a scalar beyond the narrow layout's windows: nothing of this run is usable -- the collector runs the call again
                    publish();
                    return MSM_RETRY_FULL_WIDTH;
                }
            } else {
                if (over) ctx->wide_hint.store(16);
                else if (ctx->wide_hint.load() > 0) ctx->wide_hint.fetch_sub(1);
            }
        }
        if (g.merged && h_tail[6]) {
            msm_set_err(ctx, "a scalar has bits beyond the scalar_bits these bases were prepared for (msm_amd_prepare_bases): "
                             "the table holds no window for them; prepare with a larger scalar_bits or use msm_amd_multiexp");
            publish();
            return MSM_AMD_EINVAL;
        }
    }

    // ---- 6. host combine: window sums, Horner over owned windows (top first), each weighted 2^(c*w)
    auto t0 = std::chrono::steady_clock::now();
    using HF = typename HostOf<F>::type;
    using HX = XYZZ<HF>;
    static_assert(sizeof(HX) == sizeof(X), "host twin has the same byte image");
    const HX* parts = reinterpret_cast<const HX*>(sl->h_partials);
    HX total = HX::inf();
    int prev_w = -1;   // global index of the window most recently folded in
    if (sl->plane_shift >= 0) {
        // bit planes (k_bucket_planes): partial 0 of a window weighs 1, partial 1+b weighs 2^(b + plane_shift) -- all below the window's
        // width, so one pass from the top bit down meets them in order: as many doublings as before, one addition per partial
        int pos = -1;   // bit position `total` currently stands at
        for (int wl = (int)g.Wl - 1; wl >= 0; --wl) {
            const int wglob = g.merged ? 0 : wl * (int)g.nshards + (int)g.shard;
            const int start = (int)window_start_bit(g, (uint32_t)wglob);
            for (int k = (int)sl->part_cnt[wl] - 1; k >= 0; --k) {
                const int at = start + (k == 0 ? 0 : k - 1 + sl->plane_shift);
                for (; pos > at; --pos) total = HX::dbl(total);
                pos = at;
                total = HX::add(total, parts[(size_t)sl->part_off[wl] + k]);
            }
        }
        for (; pos > 0; --pos) total = HX::dbl(total);
    } else {
    for (int wl = (int)g.Wl - 1; wl >= 0; --wl) {
        int wglob = g.merged ? 0 : wl * (int)g.nshards + (int)g.shard;   // (prepared table: the window weights are in the table)
        if (prev_w >= 0)
            for (int k = 0; k < (int)(window_start_bit(g, (uint32_t)prev_w) - window_start_bit(g, (uint32_t)wglob)); ++k) total = HX::dbl(total);
        HX s = HX::inf();
        for (uint32_t k = 0; k < sl->part_cnt[wl]; ++k) s = HX::add(s, parts[(size_t)sl->part_off[wl] + k]);
        total = HX::add(total, s);
        prev_w = wglob;
    }
    for (int k = 0; prev_w > 0 && k < (int)window_start_bit(g, (uint32_t)prev_w); ++k) total = HX::dbl(total);
    }
    if (sl->raw_out) host_store_jacobian_raw(total, out);
    else host_store_jacobian_normalised(total, out);
    auto t1 = std::chrono::steady_clock::now();

    float ms;
    (void)hipEventElapsedTime(&ms, sl->ev[0], sl->ev[7]); tm[0] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[1], sl->ev[2]); tm[1] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[2], sl->ev[3]); tm[2] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[3], sl->ev[4]); tm[3] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[9], sl->ev[5]); tm[4] = ms;
    (void)hipEventElapsedTime(&ms, sl->ev[5], sl->ev[6]); tm[5] = ms;
    tm[6] = std::chrono::duration<float, std::milli>(t1 - t0).count();
    (void)hipEventElapsedTime(&ms, sl->ev[0], sl->ev[1]); tm[7] = ms;
    publish();
    return 0;
}

