// dist.hip -- the multi-GPU exchange behind the C ABI.  The reference offers two seams, and a call is split along
// both: its per-window entry point fnName_chunk (wasmcurves/src/build_multiexp.js:96-104, :469), which its external caller fans out
// to workers, and the linearity of the MSM over slices of the points (what that caller does above 2^22 points; :251-371 is one slice).
//
// One process per GPU.  The `world` ranks form gp point slices x gw window groups (gp * gw = world, chosen per call by
// msm_dist_split below): rank r = p * gw + w computes the windows {k : k mod gw = w} of slice p of the points on its own
// MI355X (msm_amd_multiexp_shard_async), which leaves ONE Jacobian partial, already weighted by 2^(c*k).  The partials -- 144 bytes
// for BLS12-381 G1, at most 288 for G2, behind a 16-byte status header -- cross xGMI in ONE ncclAllGather (elliptic-curve addition is
// not an RCCL reduction operator, hence gather + add), and every rank adds them on the host.  Latency-bound by construction.
// A rank needs only ITS slice of the bases and scalars in HBM (msm_amd_dist_plan + msm_amd_multiexp_dist_local).
//
// RCCL is loaded at run time (dlopen of librccl.so.1): a host that never calls msm_amd_dist_init needs no RCCL at all, and a
// Python host that has torch loaded shares torch's copy instead of bringing a second one into the process.
#include <dlfcn.h>

#include "engine_ctx.hpp"

namespace {

// the handful of RCCL entry points used (signatures of rccl.h, ROCm 7.x)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void* id128) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, msm_nccl_id id, int rank) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.why = std::string("cannot load RCCL (librccl.so.1): ") + dlerror();
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
            r.why = "librccl.so lacks an expected entry point";
            r.lib = nullptr;
        }
    });
    return r;
}
constexpr int kNcclUint8 = 1;   // ncclUint8 (rccl.h)
// what one rank contributes to an exchange: a 16-byte header (word 0: 0 = fine, else the negated error code of a rank whose own part
// failed) and its Jacobian partial in 288 bytes (the largest, BLS12-381 G2).  The same size for every curve and group.
constexpr size_t kHeader = 16, kPayload = kHeader + 288;

int nccl_fail(msm_amd_ctx* ctx, const char* what, int rc) {
    msm_set_err(ctx, std::string(what) + ": " + rccl().GetErrorString(rc));
    return MSM_AMD_EHIP;
}

}  // namespace

// How a call of a multi-GPU group is split: gp point slices x gw window groups, gp * gw = world.  Both seams are the reference's own
// (per-window fnName_chunk, build_multiexp.js:96-104; the full call as a sum over point slices, :251-371, which is what its external
// caller does above 2^22 points).  Cutting the points keeps every rank's bucket count and halves the bucket populations; cutting the
// windows keeps the populations and divides the buckets -- and with them the latency-bound reduction, which is what a small call
// consists of.  The estimate below is the per-rank cost model fitted to the measurements of tools/bench_shard.py (one rank's share run
// alone on one MI355X, profiles/r03_shard_per_rank_table.txt): for every divisor gw of the world size, the time of the busiest rank.
// kShareMs[log2(points per rank) - 13][log2(window groups)]: ms per call of one rank's share -- window group 0 of gw on a slice of that
// many points, run alone on one MI355X with two, three or four calls in flight, whichever is fastest (BLS12-381 G1, 255-bit scalars;
// tools/shard_table.py, profiles/r03_shard_per_rank_table.txt; cells the sweep does not visit are geometric means of their neighbours).
// Other curves and groups scale the table; the choice of a split only needs its shape.
static const float kShareMs[14][4] = {
    {0.409f, 0.420f, 0.360f, 0.400f},       // 2^13
    {0.468f, 0.422f, 0.358f, 0.400f},       // 2^14
    {0.456f, 0.436f, 0.358f, 0.400f},       // 2^15
    {0.537f, 0.493f, 0.385f, 0.407f},       // 2^16
    {0.667f, 0.564f, 0.431f, 0.450f},       // 2^17
    {1.018f, 0.646f, 0.494f, 0.507f},       // 2^18
    {1.638f, 0.849f, 0.618f, 0.600f},       // 2^19
    {2.642f, 1.502f, 0.890f, 0.720f},       // 2^20
    {5.359f, 3.032f, 1.615f, 1.160f},       // 2^21
    {11.084f, 5.868f, 3.367f, 1.853f},      // 2^22
    {20.108f, 10.494f, 6.534f, 3.470f},     // 2^23
    {38.002f, 20.000f, 11.852f, 6.484f},    // 2^24
    {71.697f, 39.131f, 23.514f, 12.460f},   // 2^25
    {139.552f, 76.830f, 46.491f, 23.944f},  // 2^26
};
static double shard_cost_ms(uint32_t n_slice, uint32_t gw) {
    if (n_slice == 0) return 0.0;
    const double lx = std::log2((double)n_slice), x = std::min(std::max(lx - 13.0, 0.0), 13.0);
    const double y = std::min(std::max(std::log2((double)gw), 0.0), 3.0);
    const int x0 = std::min((int)x, 12), y0 = std::min((int)y, 2);
    const double fx = x - x0, fy = y - y0;
    auto at = [&](int i, int j) { return std::log((double)kShareMs[i][j]); };   // interpolated in log space
    double t = std::exp((1 - fx) * ((1 - fy) * at(x0, y0) + fy * at(x0, y0 + 1)) + fx * ((1 - fy) * at(x0 + 1, y0) + fy * at(x0 + 1, y0 + 1)));
    if (lx > 26.0) t *= (double)n_slice / 67108864.0;   // beyond the table: linear in the points
    if (gw > 8) t *= 8.0 / gw;                          // (more window groups than measured: fewer windows each)
    return t;
}
void msm_dist_split(const msm_amd_ctx* ctx, int curve, int group, uint32_t scalar_size, uint32_t n, uint32_t world, uint32_t* gp, uint32_t* gw) {
    *gp = 1;
    *gw = world ? world : 1;
    if (world <= 1) return;
    if (ctx && ctx->dist_gw && world % ctx->dist_gw == 0) {
        *gw = ctx->dist_gw;
        *gp = world / ctx->dist_gw;
        return;
    }
    (void)curve;
    (void)group;
    (void)scalar_size;
    // the split whose busiest rank -- slice 0, the longest -- is estimated fastest; near-ties (3 %) go to more point slices: the ranks
    // then hold less of the inputs
    double best = 1e300;
    for (uint32_t p = world; p >= 1; --p) {
        if (world % p) continue;
        const uint32_t w = world / p;
        const uint32_t n_slice = (uint32_t)(((uint64_t)n + p - 1) / p);
        if (p > 1 && n_slice < 1024) continue;   // slices this small are all fixed cost
        const double t = shard_cost_ms(n_slice, w);
        if (t < best * 0.97) {
            best = t;
            *gw = w;
            *gp = p;
        }
    }
}

// this context's place in its group, read under dist_mu (dist_init / dist_destroy write it under the same lock); false = no group
static bool dist_group(const msm_amd_ctx* ctx, int* rank, int* world) {
    std::lock_guard<std::mutex> dl(ctx->dist_mu);
    if (!ctx->dist.comm) return false;
    *rank = ctx->dist.rank;
    *world = ctx->dist.world;
    return true;
}

extern "C" {

int msm_amd_dist_unique_id(uint8_t* out128) {
    if (!out128) return MSM_AMD_EINVAL;
    Rccl& r = rccl();
    if (!r.lib) return MSM_AMD_EHIP;
    msm_nccl_id id;
    if (r.GetUniqueId(&id) != 0) return MSM_AMD_EHIP;
    memcpy(out128, id.internal, sizeof id.internal);
    return 0;
}

int msm_amd_dist_init(msm_amd_ctx* ctx, int rank, int world, const uint8_t* id128) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> dl(ctx->dist_mu);   // ctx->dist is guarded by dist_mu
    if (world < 1 || rank < 0 || rank >= world || !id128) {
        msm_set_err(ctx, "dist_init: need 0 <= rank < world and the 128-byte id of rank 0's msm_amd_dist_unique_id");
        return MSM_AMD_EINVAL;
    }
    if (ctx->dist.comm) {
        msm_set_err(ctx, "dist_init: this context already belongs to a group (msm_amd_dist_destroy first)");
        return MSM_AMD_EINVAL;
    }
    Rccl& r = rccl();
    if (!r.lib) {
        msm_set_err(ctx, r.why);
        return MSM_AMD_EHIP;
    }
    // Sharded calls keep up to four small accumulations in flight on four streams; with HIP's default of four hardware queues they
    // share queues with the sort streams (8-way shard, round 1: 0.82 instead of 0.58 ms per call).  The variable is read
    // when the HIP runtime starts, i.e. before this context existed: it cannot be set from here, only checked.
    if (world > 4) {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < 8)
            fprintf(stderr, "msm_amd: world = %d but GPU_MAX_HW_QUEUES is %s: export GPU_MAX_HW_QUEUES=8 before the process starts "
                            "(sharded calls in flight may share hardware queues with the sort stream and serialise)\n", world, q ? q : "unset");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // staging first, the communicator last, and ctx->dist is only written once everything exists: a failure on the way leaves the
    // context outside any group (a later dist_init may try again) and leaks nothing
    msm_amd_ctx::Dist d;
    d.rank = rank;
    d.world = world;
    d.slot_bytes = kPayload;
    const size_t cap = (size_t)(world + 1) * kPayload;
    hipError_t e = hipMalloc(&d.d_buf, cap);
    if (e == hipSuccess) e = hipHostMalloc(&d.h_buf, cap, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
    int nrc = 0;
    if (e == hipSuccess) {
        msm_nccl_id id;
        memcpy(id.internal, id128, sizeof id.internal);
        nrc = r.CommInitRank(&d.comm, world, id, rank);
    }
    if (e != hipSuccess || nrc != 0) {
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.h_buf) (void)hipHostFree(d.h_buf);
        if (d.d_buf) (void)hipFree(d.d_buf);
        if (e != hipSuccess) {
            msm_set_err(ctx, std::string("dist_init: ") + hipGetErrorString(e));
            return MSM_AMD_EHIP;
        }
        return nccl_fail(ctx, "ncclCommInitRank", nrc);
    }
    ctx->dist = std::move(d);
    return 0;
}

int msm_amd_dist_destroy(msm_amd_ctx* ctx) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> dl(ctx->dist_mu);
    if (!ctx->dist.comm) return 0;
    (void)hipSetDevice(ctx->device);
    if (ctx->dist.stream) {
        (void)hipStreamSynchronize(ctx->dist.stream);
        (void)hipStreamDestroy(ctx->dist.stream);
    }
    (void)rccl().CommDestroy(ctx->dist.comm);
    if (ctx->dist.d_buf) (void)hipFree(ctx->dist.d_buf);
    if (ctx->dist.h_buf) (void)hipHostFree(ctx->dist.h_buf);
    ctx->dist = msm_amd_ctx::Dist{};
    ctx->dist_cv.notify_all();
    return 0;
}

int msm_amd_dist_info(const msm_amd_ctx* ctx, int* rank, int* world) {
    if (!ctx || !rank || !world) return MSM_AMD_EINVAL;
    if (!dist_group(ctx, rank, world)) *rank = *world = 0;
    return 0;
}

// How a call of n points is split over a group of `world` ranks, and which slice of the points rank `rank` takes: gp point slices x
// gw window groups.  world = 0: this context's own group and rank.
int msm_amd_dist_plan(const msm_amd_ctx* ctx, int curve, int group, uint32_t scalar_size, uint32_t n, uint32_t world, uint32_t rank, uint32_t* gp,
                      uint32_t* gw, uint32_t* first, uint32_t* count) {
    if (!gp || !gw || !first || !count) return MSM_AMD_EINVAL;
    if (world == 0) {
        if (!ctx) return MSM_AMD_EINVAL;
        int r = 0, w = 0;
        if (!dist_group(ctx, &r, &w)) return MSM_AMD_EINVAL;
        world = (uint32_t)w;
        rank = (uint32_t)r;
    }
    if (rank >= world || scalar_size < 1 || scalar_size > 64) return MSM_AMD_EINVAL;
    msm_dist_split(ctx, curve, group, scalar_size, n, world, gp, gw);
    msm_dist_slice(n, *gp, rank / *gw, first, count);
    return 0;
}

}  // extern "C"

namespace {

// Queue this rank's part -- the windows {k : k % gw == w} of the `count` points behind bases / scalars -- and give it its place in the
// group's sequence of exchanges.  A part that cannot be queued (bad argument, HIP failure, no free slot) still gets a ticket and a
// place: the other ranks WILL enter the collective of this call, so this rank has to as well (with a failure mark, see dist_wait).
int dist_submit(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size, uint32_t count,
                uint32_t w, uint32_t gw, uint64_t* ticket, int refused_rc = 0, const std::string& refused_why = std::string(),
                msm_amd_bases* prepared = nullptr) {
    // (dist_mu is held across the submission so that two threads cannot take sequence numbers in one order and engine tickets in
    // the other; the submission itself only queues GPU work)
    std::lock_guard<std::mutex> dl(ctx->dist_mu);
    msm_amd_ctx::Dist::Rec rec;
    rec.curve = curve;
    rec.group = group;
    if (refused_rc) {   // the entry point has already found the call unusable on this rank
        rec.rc = refused_rc;
        rec.err = refused_why;
    } else {
        // (a prepared table holds every window of its points: such a part is a whole MSM of this rank's points)
        rec.rc = prepared ? msm_amd_multiexp_prepared_async(ctx, prepared, flags, scalars, scalar_size, count, &rec.engine_ticket)
                          : msm_amd_multiexp_shard_async(ctx, curve, group, flags, bases, scalars, scalar_size, count, w, gw, &rec.engine_ticket);
        if (rec.rc) rec.err = msm_amd_last_error(ctx);
    }
    rec.seq = ctx->dist.next_seq++;
    const uint64_t id = ctx->dist.next_id++;
    ctx->dist.recs[id] = rec;
    *ticket = id;
    return 0;
}

}  // namespace

extern "C" {

// Every rank holds ALL n points and scalars; this rank works on its slice and windows of them.
int msm_amd_multiexp_dist_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                                uint32_t n, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    int my_rank = 0, world = 0;
    if (!dist_group(ctx, &my_rank, &world)) {
        msm_set_err(ctx, "multiexp_dist: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || scalar_size < 1 || scalar_size > 64) {
        msm_set_err(ctx, "multiexp_dist: bad curve / group / scalar_size");
        return MSM_AMD_EINVAL;   // (the same on every rank: nothing was queued anywhere)
    }
    uint32_t gp, gw, first, count;
    msm_dist_split(ctx, curve, group, scalar_size, n, (uint32_t)world, &gp, &gw);
    msm_dist_slice(n, gp, (uint32_t)my_rank / gw, &first, &count);
    const size_t bstride = (size_t)((flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * (curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group;
    const uint8_t* b = bases ? (const uint8_t*)bases + (size_t)first * bstride : nullptr;
    const uint8_t* s = scalars ? (const uint8_t*)scalars + (size_t)first * scalar_size : nullptr;
    return dist_submit(ctx, curve, group, flags, b, s, scalar_size, count, (uint32_t)my_rank % gw, gw, ticket);
}

// This rank holds only ITS slice of the n_total points (msm_amd_dist_plan says which): bases_local / scalars_local are that slice.
int msm_amd_multiexp_dist_local_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases_local, const void* scalars_local,
                                      uint32_t scalar_size, uint32_t n_local, uint32_t n_total, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    int my_rank = 0, world = 0;
    if (!dist_group(ctx, &my_rank, &world)) {
        msm_set_err(ctx, "multiexp_dist: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || scalar_size < 1 || scalar_size > 64) {
        msm_set_err(ctx, "multiexp_dist: bad curve / group / scalar_size");
        return MSM_AMD_EINVAL;
    }
    uint32_t gp, gw, first, count;
    msm_dist_split(ctx, curve, group, scalar_size, n_total, (uint32_t)world, &gp, &gw);
    msm_dist_slice(n_total, gp, (uint32_t)my_rank / gw, &first, &count);
    if (n_local != count) {
        // the caller's buffers hold n_local points; reading `count` of them would run past their end (or drop points).  The split
        // depends on the context's options: plan and call must see the same ones.  Symmetric across ranks only if the caller's mistake
        // is -- so this rank still takes part in the exchange, with a failure mark.
        const std::string why = "multiexp_dist_local: n_local = " + std::to_string(n_local) + " but this rank's slice of " + std::to_string(n_total) +
                                " points is " + std::to_string(count) + " (msm_amd_dist_plan; were options changed in between?)";
        return dist_submit(ctx, curve, group, flags, nullptr, nullptr, scalar_size, count, (uint32_t)my_rank % gw, gw, ticket, MSM_AMD_EINVAL, why);
    }
    return dist_submit(ctx, curve, group, flags, bases_local, scalars_local, scalar_size, count, (uint32_t)my_rank % gw, gw, ticket);
}

// Waits for this rank's partial, takes part in the exchange of this call (one ncclAllGather of world x 304 bytes: a status word and
// the Jacobian partial of every rank) and writes the normalised sum -- the same bytes on every rank.
// The exchanges of a context happen in the order in which its calls were SUBMITTED, whatever the order (and the threads) in which they
// are waited for: every rank submits the same calls in the same order, so the collectives match -- two promises resolved by two
// libuv workers cannot cross, and a G1 call cannot meet a G2 call.  Every ticket must be waited for (a later one waits its turn).
// A rank whose own part failed still takes part, with a failure mark: it returns its error, the others MSM_AMD_EPEER.
int msm_amd_dist_wait(msm_amd_ctx* ctx, int curve, int group, uint64_t ticket, uint8_t* out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    msm_amd_ctx::Dist::Rec rec;
    {
        std::lock_guard<std::mutex> dl(ctx->dist_mu);
        if (!ctx->dist.comm) {
            msm_set_err(ctx, "dist_wait: msm_amd_dist_init first");
            return MSM_AMD_EINVAL;
        }
        auto it = ctx->dist.recs.find(ticket);
        if (it == ctx->dist.recs.end() || it->second.waiting) {
            msm_set_err(ctx, "dist_wait: unknown or already collected ticket");
            return MSM_AMD_EINVAL;
        }
        if (it->second.curve != curve || it->second.group != group) {
            msm_set_err(ctx, "dist_wait: curve / group differ from the call that issued the ticket");
            return MSM_AMD_EINVAL;
        }
        it->second.waiting = true;
        rec = it->second;
    }
    const size_t jsz = (size_t)3 * (curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group;
    uint8_t mine[kPayload] = {};
    int rc_local = rec.rc;
    std::string err_local = rec.err;
    if (rc_local == 0) {
        rc_local = msm_amd_wait(ctx, rec.engine_ticket, mine + kHeader);   // outside dist_mu: other calls are submitted and waited for meanwhile
        if (rc_local) err_local = msm_amd_last_error(ctx);
    }
    if (rc_local) {   // the point at infinity and the mark
        memset(mine, 0, sizeof mine);
        const uint32_t mark = (uint32_t)(-rc_local);
        memcpy(mine, &mark, 4);
    }
    std::unique_lock<std::mutex> dl(ctx->dist_mu);
    ctx->dist_cv.wait(dl, [&] { return !ctx->dist.comm || ctx->dist.next_exchange == rec.seq; });
    if (!ctx->dist.comm) {
        msm_set_err(ctx, "dist_wait: the group was destroyed while a call was waiting for its exchange");
        return MSM_AMD_EINVAL;
    }
    int rc = 0;
    const int world = ctx->dist.world;
    uint8_t* d = (uint8_t*)ctx->dist.d_buf;
    uint8_t* h = (uint8_t*)ctx->dist.h_buf;
    do {
        if (hipSetDevice(ctx->device) != hipSuccess) { rc = MSM_AMD_EHIP; msm_set_err(ctx, "dist_wait: hipSetDevice failed"); break; }
        memcpy(h, mine, kPayload);
        hipError_t e = hipMemcpyAsync(d, h, kPayload, hipMemcpyHostToDevice, ctx->dist.stream);
        int nrc = 0;
        if (e == hipSuccess) nrc = rccl().AllGather(d, d + kPayload, kPayload, kNcclUint8, ctx->dist.comm, ctx->dist.stream);
        if (e == hipSuccess && nrc == 0) e = hipMemcpyAsync(h + kPayload, d + kPayload, kPayload * (size_t)world, hipMemcpyDeviceToHost, ctx->dist.stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->dist.stream);
        if (nrc != 0) { rc = nccl_fail(ctx, "ncclAllGather", nrc); break; }
        if (e != hipSuccess || e2 != hipSuccess) {
            rc = MSM_AMD_EHIP;
            msm_set_err(ctx, std::string("dist_wait: ") + hipGetErrorString(e != hipSuccess ? e : e2));
            break;
        }
    } while (0);
    // this call's turn is over whatever happened: the next one may go
    ++ctx->dist.next_exchange;
    ctx->dist.recs.erase(ticket);
    std::vector<uint8_t> parts;
    int failed_rank = -1;
    if (rc == 0) {
        parts.resize((size_t)world * jsz);
        for (int k = 0; k < world; ++k) {
            uint32_t mark;
            memcpy(&mark, h + kPayload * (size_t)(k + 1), 4);
            if (mark && failed_rank < 0) failed_rank = k;
            memcpy(parts.data() + (size_t)k * jsz, h + kPayload * (size_t)(k + 1) + kHeader, jsz);
        }
    }
    dl.unlock();
    ctx->dist_cv.notify_all();
    if (rc) return rc;
    if (rc_local) {
        msm_set_err(ctx, err_local);
        return rc_local;
    }
    if (failed_rank >= 0) {
        msm_set_err(ctx, "multiexp_dist: rank " + std::to_string(failed_rank) + " of the group failed its part of this call (its own error text says why)");
        return MSM_AMD_EPEER;
    }
    return msm_amd_combine(curve, group, parts.data(), (uint32_t)world, out);
}

int msm_amd_multiexp_dist(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                          uint32_t n, uint8_t* out) {
    uint64_t t = 0;
    if (int rc = msm_amd_multiexp_dist_async(ctx, curve, group, flags, bases, scalars, scalar_size, n, &t)) return rc;
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

// The points of the group cut along the points only, every rank holding the PREPARED table of its own slice (msm_amd_prepare_bases on
// whatever slice the host gave it: the slices need not be equal): this rank's part is msm_amd_multiexp_prepared over its n_local points.
int msm_amd_multiexp_dist_prepared_async(msm_amd_ctx* ctx, msm_amd_bases* local_table, uint32_t flags, const void* scalars_local, uint32_t scalar_size,
                                         uint32_t n_local, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    int my_rank = 0, world = 0;
    if (!dist_group(ctx, &my_rank, &world)) {
        msm_set_err(ctx, "multiexp_dist: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    (void)my_rank;
    (void)world;
    uint32_t info[6];
    if (!local_table || msm_amd_bases_info(local_table, info, nullptr)) {
        msm_set_err(ctx, "multiexp_dist_prepared: no table");
        return MSM_AMD_EINVAL;
    }
    int curve = 0, group = 0;
    msm_bases_kind(local_table, &curve, &group);
    return dist_submit(ctx, curve, group, flags, nullptr, scalars_local, scalar_size, n_local, 0, 1, ticket, 0, std::string(), local_table);
}
int msm_amd_multiexp_dist_prepared(msm_amd_ctx* ctx, msm_amd_bases* local_table, uint32_t flags, const void* scalars_local, uint32_t scalar_size,
                                   uint32_t n_local, uint8_t* out) {
    uint64_t t = 0;
    if (int rc = msm_amd_multiexp_dist_prepared_async(ctx, local_table, flags, scalars_local, scalar_size, n_local, &t)) return rc;
    int curve = 0, group = 0;
    msm_bases_kind(local_table, &curve, &group);
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

int msm_amd_multiexp_dist_local(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases_local, const void* scalars_local,
                                uint32_t scalar_size, uint32_t n_local, uint32_t n_total, uint8_t* out) {
    uint64_t t = 0;
    if (int rc = msm_amd_multiexp_dist_local_async(ctx, curve, group, flags, bases_local, scalars_local, scalar_size, n_local, n_total, &t)) return rc;
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

}  // extern "C"
