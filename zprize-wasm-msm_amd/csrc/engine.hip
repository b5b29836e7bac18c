// engine.hip -- MI355X (gfx950) Pippenger multi-scalar multiplication behind the C ABI of
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
#include "engine_ctx.hpp"
#include "sort_kernels.hpp"

// msm_amd_last_error: a per-thread copy taken under the error lock (the stored string may be rewritten by another thread)
static std::mutex g_err_mu;
void msm_set_err(msm_amd_ctx* ctx, const std::string& msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    ctx->err = msg;
}

// Carries exactly MSM_SCRATCH_RESERVE bytes of private segment per lane; dispatched (one empty wave) on every stream of a new context so
// that the runtime sizes every queue's scratch while a failure can still be reported (engine_ctx.hpp).
__global__ void k_scratch_reserve(uint32_t* out, uint32_t n, uint32_t sel) {
    if (n == 0) return;
    constexpr int WORDS = (int)(MSM_SCRATCH_RESERVE / 4);
    volatile uint32_t buf[WORDS];
    for (int i = 0; i < WORDS; ++i) buf[i] = i * sel + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < n; ++i) acc += buf[(i * sel + acc) % WORDS];
    out[threadIdx.x] = acc;
}

int msm_scratch_guard(msm_amd_ctx* ctx, int slot, const void* kernel, const char* what) {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, kernel) != hipSuccess) {
        (void)hipGetLastError();
        return 0;   // (no attributes, no verdict: launch as before)
    }
    const size_t need = at.localSizeBytes;
    if (need <= ctx->scratch_level[slot].load()) return 0;   // the queue holds that much already: the runtime keeps what it has allocated
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (free_b < need * MSM_SCRATCH_SLOTS + MSM_SCRATCH_MARGIN) {
        msm_set_err(ctx, std::string(what) + ": " + std::to_string(need) + " bytes of private segment per lane need " + std::to_string((need * MSM_SCRATCH_SLOTS) >> 20) +
                             " MB of scratch on the queue, and the device has " + std::to_string(free_b >> 20) + " MB free (out of device memory)");
        return MSM_AMD_ENOMEM;
    }
    ctx->scratch_level[slot].store(need);
    return 0;
}


#include "engine_call.hpp"

static int check_args(msm_amd_ctx* ctx, const RunArgs& a) {
    if (a.n && (!a.bases || !a.scalars)) {
        msm_set_err(ctx, "null pointer");
        return MSM_AMD_EINVAL;
    }
    if (a.scalar_size < 1 || a.scalar_size > 64) {
        msm_set_err(ctx, "scalar_size must be in [1, 64]");
        return MSM_AMD_EINVAL;
    }
    if (a.n > (1u << 27)) {   // (pair, window) positions are 32-bit: 17 windows x n must stay below 2^32
        msm_set_err(ctx, "n must not exceed 2^27 per asynchronous call (the synchronous entry points slice larger inputs themselves)");
        return MSM_AMD_EINVAL;
    }
    if (a.nshards == 0 || a.shard >= a.nshards) {
        msm_set_err(ctx, "bad shard");
        return MSM_AMD_EINVAL;
    }
    return 0;
}

// Queue one MSM on the next slot; the caller collects it with finish_ticket().
static int submit(msm_amd_ctx* ctx, int curve, int group, const RunArgs& a, uint64_t* ticket, bool wait_for_slot = false) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::unique_lock<std::mutex> lk(ctx->mu);
    if (int rc = check_args(ctx, a)) return rc;
    // (a prepared table is checked and counted in under this lock, the one msm_amd_release_bases frees it under)
    if (a.prepared && std::find(ctx->prepared_handles.begin(), ctx->prepared_handles.end(), a.prepared) == ctx->prepared_handles.end()) {
        msm_set_err(ctx, "multiexp_prepared: these bases have been released");
        return MSM_AMD_EINVAL;
    }
    Slot* sl = nullptr;
    for (;;) {
        for (uint32_t k = 0; k < ctx->nslots; ++k) {   // tickets may be collected in any order: take any free slot
            Slot* cand = &ctx->slots[(ctx->next_slot + k) % ctx->nslots];
            if (!cand->pending) {
                sl = cand;
                break;
            }
        }
        if (sl) break;
        // every slot is taken.  Slots held by synchronous calls of other threads will be collected by those threads: wait for one
        // (asynchronous submissions too: round 3 -- a thread that pipelines its own calls beside other threads' synchronous ones
        // used to be refused here).  If all of them hold asynchronous tickets, only the caller can free one -- waiting would never end.
        bool some_sync = false;
        for (uint32_t k = 0; k < ctx->nslots; ++k) some_sync |= ctx->slots[k].sync_owned;
        if (!some_sync) break;
        ctx->cv.wait(lk);
    }
    if (!sl) {
        msm_set_err(ctx, "every slot of this context holds an uncollected call: wait for an older ticket first");
        return MSM_AMD_EINVAL;
    }
    // reductions that were left for "the next thing that happens on the context" (engine_call.hpp, submit_msm): this submission is that
    // thing -- they go first, in the form that hides behind an accumulation.  A failure is kept for the wait on that ticket.
    for (uint32_t k = 0; k < ctx->nslots; ++k) {
        Slot* o = &ctx->slots[k];
        if (o != sl && o->pending && o->deferred_reduce)
            if (int drc = flush_deferred_reduce(ctx, o, true)) o->deferred_rc = drc;
    }
    sl->deferred_rc = 0;
    {
        RunArgs kept = a;       // (`a` may be this very slot's copy: the collector's second run of a call)
        sl->retry_args = kept;  // the slot's copy outlives the submission: a deferred reduction reads it
    }
    const RunArgs& as = sl->retry_args;
    int rc;
    if (curve == MSM_AMD_BLS12381 && group == 1) rc = submit_msm<Fp<Bls12381Fp>>(ctx, sl, as);
    else if (curve == MSM_AMD_BLS12381 && group == 2) rc = submit_msm<Fp2<Bls12381Fp>>(ctx, sl, as);
    else if (curve == MSM_AMD_BN128 && group == 1) rc = submit_msm<Fp<Bn254Fp>>(ctx, sl, as);
    else if (curve == MSM_AMD_BN128 && group == 2) rc = submit_msm<Fp2<Bn254Fp>>(ctx, sl, as);
    else {
        msm_set_err(ctx, "unknown curve/group");
        return MSM_AMD_EINVAL;
    }
    if (rc) {
        sl->deferred_reduce = nullptr;
        sl->binhist_zero.clear();   // whatever the sort had counted so far is not to be trusted
        sl->ctl_zero.clear();
        sl->flags_zero.clear();
        // leave the streams clean so the context stays usable
        (void)hipStreamSynchronize(ctx->stream_s);
        if (ctx->stream_s2) (void)hipStreamSynchronize(ctx->stream_s2);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(sl->stream_c);
        (void)hipStreamSynchronize(sl->stream_b);
        return rc;
    }
    sl->pending = true;
    sl->retry_curve = curve;
    sl->retry_group = group;
    sl->prepared = a.prepared;
    if (a.prepared) a.prepared->in_flight.fetch_add(1);
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
            msm_set_err(ctx, "unknown or already collected ticket");
            return MSM_AMD_EINVAL;
        }
        // nothing was submitted after this call: its reduction is queued now, in the form the moment asks for
        if (found->deferred_reduce) {
            bool others = false;
            for (Slot& sl : ctx->slots) others |= (&sl != found && sl.pending);
            found->deferred_rc = flush_deferred_reduce(ctx, found, others);
        }
        if (found->deferred_rc) {   // the reduction could not be queued (out of memory, a HIP failure): the call is over, the slot free
            const int drc = found->deferred_rc;
            found->deferred_rc = 0;
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(found->stream_c);
            (void)hipStreamSynchronize(found->stream_b);
            found->ctl_zero.clear();
            found->flags_zero.clear();
            if (found->prepared) {
                found->prepared->in_flight.fetch_sub(1);
                found->prepared = nullptr;
            }
            found->pending = false;
            found->collecting = false;
            found->sync_owned = false;
            ctx->cv.notify_all();
            return drc;
        }
    }
    // wait for the GPU outside the lock (other threads may submit and collect meanwhile)
    if (!found->trivial) {
        (void)hipSetDevice(ctx->device);
        (void)hipEventSynchronize(found->ev[7]);
    }
    // the host combine runs outside the lock (the slot stays this thread's while `collecting` is set); only the bookkeeping is locked
    int rc = found->finish(ctx, found, out);
    RunArgs again;
    int again_curve = 0, again_group = 0;
    const bool retry = rc == MSM_RETRY_FULL_WIDTH;
    if (retry) {
        again = found->retry_args;
        again.full_width = true;
        again_curve = found->retry_curve;
        again_group = found->retry_group;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (found->prepared) {
            found->prepared->in_flight.fetch_sub(1);
            found->prepared = nullptr;
        }
        found->pending = false;
        found->collecting = false;
        found->sync_owned = false;
    }
    ctx->cv.notify_all();
    if (retry) {
        // the call was planned for scalars narrower than it met (engine_call.hpp, narrow layout): once more, over the declared width --
        // the caller's buffers are still valid (they must be until the ticket has been waited for), and a slot is free: this one
        uint64_t t2 = 0;
        rc = submit(ctx, again_curve, again_group, again, &t2, true);
        if (rc == 0) rc = finish_ticket(ctx, t2, out);
    }
    return rc;
}

bool msm_ticket_geom(msm_amd_ctx* ctx, uint64_t ticket, Geom* g) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (Slot& sl : ctx->slots)
        if (sl.pending && sl.ticket == ticket) {
            if (sl.trivial) return false;
            *g = sl.g;
            return true;
        }
    return false;
}

static size_t elem_bytes(int curve, int group) { return (size_t)(curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group; }
static int dispatch(msm_amd_ctx* ctx, int curve, int group, const RunArgs& a, uint8_t* out) {
    if (!ctx) return MSM_AMD_EINVAL;
    if (!out) {
        msm_set_err(ctx, "null pointer");
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
        msm_set_err(ctx, "bad argument");
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
    if (const char* e = getenv("MSM_AMD_REDUCE_L")) {
        const uint32_t v = (uint32_t)atoi(e);
        if ((v & (v - 1)) == 0) ctx->force_L = v;   // powers of two only (see msm_amd_set_option "reduce_l")
    }
    if (const char* e = getenv("MSM_AMD_SEG")) ctx->force_seg = (uint32_t)atoi(e);
    if (const char* e = getenv("MSM_AMD_C")) ctx->force_c = (uint32_t)atoi(e);
    if (const char* e = getenv("MSM_AMD_DIST_TIMEOUT_MS")) ctx->dist_timeout_ms.store((uint32_t)strtoul(e, nullptr, 10));
    // (experiments: MSM_AMD_S_PRIORITY / MSM_AMD_A_PRIORITY create the sort stream / the accumulation stream at that HIP priority, -1 = greatest)
    auto make_stream = [](hipStream_t* st, const char* env) {
        const char* e = getenv(env);
        return e ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, atoi(e)) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    if (hipSetDevice(device_id) != hipSuccess || make_stream(&ctx->stream, "MSM_AMD_A_PRIORITY") != hipSuccess ||
        make_stream(&ctx->stream_s, "MSM_AMD_S_PRIORITY") != hipSuccess) {
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
    // scratch for the MSM path's kernels on every queue, now -- or the create fails cleanly (engine_ctx.hpp: a nearly full device would
    // otherwise abort the process at some later dispatch).  HIP maps the streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default).
    {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        const size_t queues = std::min<size_t>(2 + 2 * msm_amd_ctx::MAX_SLOTS, q && atoi(q) > 0 ? (size_t)atoi(q) : 4);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < queues * MSM_SCRATCH_RESERVE * MSM_SCRATCH_SLOTS + MSM_SCRATCH_MARGIN) {
            (void)hipGetLastError();
            fprintf(stderr, "msm_amd_create: %zu MB of device memory free, %zu MB needed for the kernels' scratch on %zu hardware queues\n", free_b >> 20,
                    (queues * MSM_SCRATCH_RESERVE * MSM_SCRATCH_SLOTS + MSM_SCRATCH_MARGIN) >> 20, queues);
            msm_amd_destroy(ctx);
            return MSM_AMD_ENOMEM;
        }
        std::vector<hipStream_t> all = {ctx->stream, ctx->stream_s};
        for (Slot& sl : ctx->slots) {
            all.push_back(sl.stream_b);
            all.push_back(sl.stream_c);
        }
        for (hipStream_t st : all) k_scratch_reserve<<<1, 64, 0, st>>>(nullptr, 0, 1);
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
            msm_amd_destroy(ctx);
            return MSM_AMD_EHIP;
        }
        ctx->scratch_level[0].store(MSM_SCRATCH_RESERVE);
        ctx->scratch_level[1].store(MSM_SCRATCH_RESERVE);
    }
    calibrate_streams(ctx);
    *out_ctx = ctx;
    return MSM_AMD_OK;
}

void msm_amd_destroy(msm_amd_ctx* ctx) {
    if (!ctx) return;
    (void)msm_amd_dist_destroy(ctx);
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
    for (msm_amd_bases* h : ctx->prepared_handles) {   // tables the caller never released
        if (h->table) (void)hipFree(h->table);
        delete h;
    }
    ctx->prepared_handles.clear();
    if (ctx->stream_s) (void)hipStreamDestroy(ctx->stream_s);
    if (ctx->stream_s2) (void)hipStreamDestroy(ctx->stream_s2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* msm_amd_last_error(const msm_amd_ctx* ctx) {
    if (!ctx) return "null context";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = ctx->err;
    return copy.c_str();
}

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
    std::lock_guard<std::mutex> lk(ctx->mu);   // options are read while a call is planned (submit holds the same lock)
    if (!strcmp(name, "segment")) { ctx->force_seg = value; return 0; }
    if (!strcmp(name, "reduce_l")) {
        // lanes per workgroup follow from L (rblock = min(256, max(64, buckets / L))) and the workgroup's LDS tree halves: a value that is
        // not a power of two would give an odd lane count somewhere on the way down and drop a partial sum
        if (value & (value - 1)) { msm_set_err(ctx, "reduce_l must be 0 (automatic) or a power of two"); return MSM_AMD_EINVAL; }
        ctx->force_L = value;
        return 0;
    }
    if (!strcmp(name, "defer_reduce")) { ctx->defer_reduce = value > 2 ? 2 : (int)value; return 0; }
    if (!strcmp(name, "reduce_planes")) { ctx->reduce_planes = value > 2 ? 2 : (int)value; return 0; }
    if (!strcmp(name, "split")) { ctx->no_split = value == 0; return 0; }
    if (!strcmp(name, "reduce_pair")) { ctx->reduce_pair = value != 0; return 0; }
    if (!strcmp(name, "sort_big")) { ctx->sort_big = value > 2 ? 0 : (int)value; return 0; }
    if (!strcmp(name, "glv")) { ctx->glv_mode = value > 2 ? 2 : (int)value; return 0; }
    if (!strcmp(name, "split2")) { ctx->no_split2 = value == 0; return 0; }
    if (!strcmp(name, "splitk")) {
        if (value > MAX_SLICES) return MSM_AMD_EINVAL;
        ctx->splitk = value;
        return 0;
    }
    if (!strcmp(name, "items")) { ctx->items_mode = value > 2 ? 0 : (int)value; return 0; }   // the accumulation's work list: 0 = where a launch is short of filling the chip, 1 = wherever the populations allow, 2 = never
    if (!strcmp(name, "items_t")) { ctx->items_t = value; return 0; }                         // experiments: entries per lane of the work list (0 = chosen per launch)
    if (!strcmp(name, "narrow")) {   // 0 = windows never adapt to the observed scalar width; setting it (to anything) forgets what has been observed
        ctx->narrow_mode = value != 0;
        std::lock_guard<std::mutex> hl(ctx->hint_mu);
        ctx->narrow_bits.store(0);
        ctx->width_count = 0;
        return 0;
    }
    if (!strcmp(name, "hot")) { ctx->no_hot = value == 0; return 0; }   // 0 = hot buckets take the ordinary segment path (before / after measurements)
    if (!strcmp(name, "zero_copy")) { ctx->zero_copy = value != 0; return 0; }
    if (!strcmp(name, "even_windows")) { ctx->even_windows = value != 0; return 0; }   // 0 = the round-2 layout (c-bit windows only); default 1
    if (!strcmp(name, "dist_timeout_ms")) { ctx->dist_timeout_ms.store(value); return 0; }   // how long an exchange of a multi-GPU group waits for its peers (0 = for ever)
    if (!strcmp(name, "dist_gw")) { ctx->dist_gw = value; return 0; }   // window groups of a sharded call of a multi-GPU group (0 = chosen per call)
    if (!strcmp(name, "sort_geom")) { ctx->sort_geom = value; return 0; }       // experiments / tests: 0 = round 4's geometry by size, 1 = round 3's (256 coarse bins, 9216-entry spans), 2 = 512 bins wherever the window allows, 3 = 2 + the 1024-lane fine sort
    if (!strcmp(name, "chunk_mode")) { ctx->chunk_mode = value; return 0; }     // experiments: chunk-parallel fine sort, 1 = 4096-entry chunks / 256 lanes, 2 = 8192 / 1024, 0 = by what else is in flight
    if (!strcmp(name, "tile_mode")) { ctx->tile_mode = value; return 0; }       // experiments: 1 = tiles of 1024 scalars, 2 = of 4096, 0 = by size
    if (!strcmp(name, "acc_stream")) { ctx->acc_stream = value; return 0; }     // experiments: 1 = the slot's stream, 2 = the context's stream A
    if (!strcmp(name, "acc_variant")) { ctx->acc_variant = value; return 0; }   // experiments: 1 = two-chain build, 2 = chained build, 0 = by launch size
    if (!strcmp(name, "splitk_lanes")) { ctx->splitk_lanes = value ? value : 5u * 1024u * 64u; return 0; }
    if (!strcmp(name, "slice")) {
        if (value < 1 || value > (1u << 27)) { msm_set_err(ctx, "slice must be in [1, 2^27]"); return MSM_AMD_EINVAL; }
        ctx->max_slice = value;
        return 0;
    }
    if (!strcmp(name, "slots")) {
        if (value < 1 || value > msm_amd_ctx::MAX_SLOTS) { msm_set_err(ctx, "slots must be in [1, 8]"); return MSM_AMD_EINVAL; }
        for (Slot& sl : ctx->slots)
            if (sl.pending) { msm_set_err(ctx, "cannot change the slot count with calls in flight"); return MSM_AMD_EINVAL; }
        ctx->nslots = value;
        ctx->next_slot = 0;
        return 0;
    }
    msm_set_err(ctx, "unknown option");
    return MSM_AMD_EINVAL;
}

int msm_amd_set_window_bits(msm_amd_ctx* ctx, uint32_t c) {
    if (!ctx || c > (uint32_t)MAX_WINDOW_BITS) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
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
        msm_set_err(ctx, "chunk_size must be in [1, 25]");
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

// ---- prepared bases: the table is built in prepared.hip; the calls on a handle are ordinary submissions ---------------------------
// The handle is looked up among the context's registered tables BEFORE it is dereferenced, under the lock msm_amd_release_bases takes:
// a handle of another context, or one already released (by this or another thread), is refused instead of read.  submit() repeats the
// look-up in the critical section that counts the call in, so a release between here and there is caught as well.
static int prepared_args(msm_amd_ctx* ctx, msm_amd_bases* h, uint32_t flags, const void* scalars, uint32_t scalar_size, uint32_t n, RunArgs* a, int* curve, int* group) {
    if (!ctx || !h) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (std::find(ctx->prepared_handles.begin(), ctx->prepared_handles.end(), h) == ctx->prepared_handles.end()) {
        msm_set_err(ctx, "multiexp_prepared: not bases of this context (prepared on another context, or already released)");
        return MSM_AMD_EINVAL;
    }
    if (n > h->n) {
        msm_set_err(ctx, "multiexp_prepared: n exceeds the number of prepared points");
        return MSM_AMD_EINVAL;
    }
    if (flags & ~(uint32_t)MSM_AMD_DEVICE_PTRS) {
        msm_set_err(ctx, "multiexp_prepared: the only flag is MSM_AMD_DEVICE_PTRS (scalars resident on the GPU)");
        return MSM_AMD_EINVAL;
    }
    *a = RunArgs{h->table, scalars, scalar_size, n, flags, 0, scalar_size * 8, 0, 1, false, h};
    *curve = h->curve;
    *group = h->group;
    return 0;
}
// R = sum_{i < n} scalar_i * P_i over the first n prepared points
int msm_amd_multiexp_prepared(msm_amd_ctx* ctx, msm_amd_bases* h, uint32_t flags, const void* scalars, uint32_t scalar_size, uint32_t n, uint8_t* out) {
    RunArgs a;
    int curve = 0, group = 0;
    if (int rc = prepared_args(ctx, h, flags, scalars, scalar_size, n, &a, &curve, &group)) return rc;
    if (!out) return MSM_AMD_EINVAL;
    uint64_t t = 0;
    if (int rc = submit(ctx, curve, group, a, &t, true)) return rc;
    return finish_ticket(ctx, t, out);
}
int msm_amd_multiexp_prepared_async(msm_amd_ctx* ctx, msm_amd_bases* h, uint32_t flags, const void* scalars, uint32_t scalar_size, uint32_t n,
                                    uint64_t* ticket) {
    RunArgs a;
    int curve = 0, group = 0;
    if (!ticket) return MSM_AMD_EINVAL;
    if (int rc = prepared_args(ctx, h, flags, scalars, scalar_size, n, &a, &curve, &group)) return rc;
    return submit(ctx, curve, group, a, ticket);
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

int msm_amd_get_timings(const msm_amd_ctx* ctx, float* ms, int capacity) {
    if (!ctx || !ms) return MSM_AMD_EINVAL;
    int k = std::min(capacity, 8);
    for (int i = 0; i < k; ++i) ms[i] = ctx->timings[i];
    return k;
}
int msm_amd_get_hints(const msm_amd_ctx* ctx, uint32_t* out4) {
    if (!ctx || !out4) return MSM_AMD_EINVAL;
    out4[0] = (uint32_t)std::max(0, ctx->wide_hint.load());
    out4[1] = (uint32_t)std::max(0, ctx->skew_hint.load());
    out4[2] = ctx->nslots | (std::min<uint32_t>(ctx->fill_mean16.load() / 16u, 0xffffu) << 16);
    out4[3] = ctx->last_items_T | ((uint32_t)std::max(0, ctx->narrow_bits.load()) << 16);
    return 0;
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
