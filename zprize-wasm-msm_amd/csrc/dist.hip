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

#include <thread>

#include "engine_ctx.hpp"

namespace {

// the handful of RCCL entry points used (signatures of rccl.h, ROCm 7.x)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void* id128) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, msm_nccl_id id, int rank) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*CommAbort)(void* comm) = nullptr;
    int (*CommCount)(const void* comm, int* count) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why, name;
};
// MSM_AMD_RCCL_LIB names the library that provides the collective (default: librccl.so.1).  The tests point it at a stand-in that
// carries the same five calls over POSIX shared memory (tests/host/fake_rccl.c), so that the multi-rank code of this file runs with
// two real processes on a one-GPU box, which RCCL itself refuses (two ranks on one device).
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* forced = getenv("MSM_AMD_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (forced && *forced) {
            r.lib = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
            r.name = forced;
        } else {
            for (const char* n : names) {
                r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                r.name = n;
                if (r.lib) break;
            }
        }
        if (!r.lib) {
            const char* de = dlerror();
            r.why = std::string("cannot load the collective library (") + (forced && *forced ? forced : "librccl.so.1") + "): " + (de ? de : "?");
            return;
        }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.CommAbort || !r.GetErrorString) {
            r.why = r.name + " lacks an expected entry point";
            r.lib = nullptr;
        }
    });
    return r;
}
constexpr int kNcclUint8 = 1;   // ncclUint8 (rccl.h)
// what one rank contributes to an exchange: a 16-byte header and its Jacobian partial in 288 bytes (the largest, BLS12-381 G2).  The
// same size for every curve and group.  Header: word 0: 0 = fine, else the negated error code of a rank whose own part failed;
// word 1: the split this rank planned, gp | gw << 16; word 2: the total number of points it was told; word 3: a hash of the window
// layout of its part (0 = nothing to compare) -- Dist::Rec in engine_ctx.hpp.
constexpr size_t kHeader = 16, kPayload = kHeader + 288;

uint32_t layout_hash(int curve, int group, uint32_t scalar_size, const Geom& g) {
    uint32_t h = 2166136261u;
    for (uint32_t v : {(uint32_t)curve, (uint32_t)group, scalar_size, g.c, g.W, g.ns, g.nshards, g.merged}) h = (h ^ v) * 16777619u;
    return h ? h : 1u;
}

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
// tools/shard_table.py, profiles/r04_shard_per_rank_table.txt -- re-measured in round 4 with the work list and the plane kernels; tools/fit_share_tables.py: cells the
// sweep does not visit keep the round-3 value scaled by the nearest re-measured cell).
// Other curves and groups scale the table; the choice of a split only needs its shape.
static const float kShareMsG1[14][4] = {
    {0.391f, 0.401f, 0.344f, 0.382f},   // 2^13  (partly scaled from the G1 table)
    {0.444f, 0.401f, 0.340f, 0.380f},   // 2^14  (partly scaled from the G1 table)
    {0.319f, 0.384f, 0.284f, 0.317f},   // 2^15  (partly scaled from the G1 table)
    {0.414f, 0.384f, 0.297f, 0.357f},   // 2^16
    {0.589f, 0.406f, 0.314f, 0.328f},   // 2^17  (partly scaled from the G1 table)
    {0.906f, 0.550f, 0.384f, 0.393f},   // 2^18
    {1.464f, 0.907f, 0.552f, 0.536f},   // 2^19  (partly scaled from the G1 table)
    {2.561f, 1.477f, 0.861f, 0.699f},   // 2^20
    {5.070f, 3.053f, 1.587f, 1.140f},   // 2^21  (partly scaled from the G1 table)
    {10.208f, 6.297f, 3.032f, 1.784f},   // 2^22
    {18.744f, 9.651f, 5.920f, 3.143f},   // 2^23  (partly scaled from the G1 table)
    {35.458f, 18.595f, 11.973f, 6.127f},   // 2^24
    {67.586f, 37.081f, 23.423f, 12.412f},   // 2^25  (partly scaled from the G1 table)
    {130.893f, 73.006f, 44.413f, 23.578f},   // 2^26
};
// The same for BN254 G1 and for G2 (round 4; tools/shard_table.py --curve bn128 / --group 2, profiles/r04_shard_table_*.txt, built with
// tools/fit_share_tables.py: a cell is the best of two, three and four calls in flight; cells the sweeps did not visit are the G1 table's
// scaled by the nearest measured cell).  BN254's cheaper field moves its cross-over between point slices and window groups to larger
// shares than BLS12-381's; G2 (measured on BLS12-381, used for both curves' G2: the split only needs the table's shape) costs ~3x per point.
static const float kShareMsBn254G1[14][4] = {
    {0.246f, 0.253f, 0.217f, 0.241f},   // 2^13  (partly scaled from the G1 table)
    {0.280f, 0.322f, 0.273f, 0.305f},   // 2^14  (partly scaled from the G1 table)
    {0.286f, 0.325f, 0.257f, 0.287f},   // 2^15  (partly scaled from the G1 table)
    {0.311f, 0.345f, 0.273f, 0.318f},   // 2^16
    {0.364f, 0.378f, 0.297f, 0.310f},   // 2^17  (partly scaled from the G1 table)
    {0.504f, 0.388f, 0.324f, 0.340f},   // 2^18
    {0.839f, 0.471f, 0.343f, 0.333f},   // 2^19  (partly scaled from the G1 table)
    {1.277f, 0.773f, 0.481f, 0.428f},   // 2^20
    {2.534f, 1.502f, 0.870f, 0.625f},   // 2^21  (partly scaled from the G1 table)
    {5.081f, 3.286f, 1.632f, 0.854f},   // 2^22
    {10.511f, 5.480f, 3.363f, 1.786f},   // 2^23  (partly scaled from the G1 table)
    {19.678f, 10.514f, 6.851f, 3.383f},   // 2^24
    {37.126f, 20.572f, 13.592f, 6.502f},   // 2^25  (partly scaled from the G1 table)
    {72.263f, 40.391f, 26.874f, 12.494f},   // 2^26  (partly scaled from the G1 table)
};
static const float kShareMsG2[14][4] = {
    {0.464f, 0.477f, 0.408f, 0.454f},   // 2^13  (partly scaled from the G1 table)
    {0.693f, 0.789f, 0.669f, 0.748f},   // 2^14  (partly scaled from the G1 table)
    {0.915f, 0.810f, 0.596f, 0.666f},   // 2^15  (partly scaled from the G1 table)
    {1.075f, 0.862f, 0.614f, 0.452f},   // 2^16
    {1.565f, 1.037f, 0.683f, 0.713f},   // 2^17  (partly scaled from the G1 table)
    {2.655f, 1.476f, 0.875f, 0.632f},   // 2^18
    {4.410f, 2.367f, 1.346f, 1.307f},   // 2^19  (partly scaled from the G1 table)
    {7.983f, 4.193f, 2.242f, 1.349f},   // 2^20
    {15.210f, 8.047f, 4.213f, 3.026f},   // 2^21  (partly scaled from the G1 table)
    {30.089f, 15.344f, 8.114f, 4.718f},   // 2^22
    {54.587f, 27.441f, 15.745f, 8.835f},   // 2^23  (partly scaled from the G1 table)
    {103.163f, 52.298f, 28.560f, 16.508f},   // 2^24  (partly scaled from the G1 table)
    {194.634f, 102.323f, 56.663f, 31.723f},   // 2^25  (partly scaled from the G1 table)
    {378.837f, 200.901f, 112.032f, 60.961f},   // 2^26  (partly scaled from the G1 table)
};
static double shard_cost_ms(const float (*kShareMs)[4], uint32_t n_slice, uint32_t gw) {
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
    (void)scalar_size;
    const float (*table)[4] = group == 2 ? kShareMsG2 : (curve == MSM_AMD_BN128 ? kShareMsBn254G1 : kShareMsG1);
    // the split whose busiest rank -- slice 0, the longest -- is estimated fastest; near-ties (3 %) go to more point slices: the ranks
    // then hold less of the inputs
    double best = 1e300;
    for (uint32_t p = world; p >= 1; --p) {
        if (world % p) continue;
        const uint32_t w = world / p;
        const uint32_t n_slice = (uint32_t)(((uint64_t)n + p - 1) / p);
        if (p > 1 && n_slice < 1024) continue;   // slices this small are all fixed cost
        const double t = shard_cost_ms(table, n_slice, w);
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
// the same for an entry point that is about to queue work: a dead group refuses at once (MSM_AMD_EPEER)
static int dist_group_alive(msm_amd_ctx* ctx, int* rank, int* world) {
    std::string why;
    {
        std::lock_guard<std::mutex> dl(ctx->dist_mu);
        if (!ctx->dist.comm) {
            why = "multiexp_dist: msm_amd_dist_init first";
        } else if (ctx->dist.dead) {
            why = "multiexp_dist: this group is dead (" + ctx->dist.dead_why + "): msm_amd_dist_destroy, then a new msm_amd_dist_init on every rank";
        } else {
            *rank = ctx->dist.rank;
            *world = ctx->dist.world;
            return 0;
        }
    }
    const bool dead = why.find("dead") != std::string::npos;
    msm_set_err(ctx, why);
    return dead ? MSM_AMD_EPEER : MSM_AMD_EINVAL;
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
    if (world > 65535) {
        msm_set_err(ctx, "dist_init: at most 65535 ranks");
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
            (void)hipGetLastError();
            return e == hipErrorOutOfMemory ? MSM_AMD_ENOMEM : MSM_AMD_EHIP;
        }
        return nccl_fail(ctx, "ncclCommInitRank", nrc);
    }
    ctx->dist = std::move(d);
    ++ctx->dist_epoch;
    return 0;
}

// Leaves the group.  Calls of the group that were submitted and never waited for are collected here (their slots would otherwise stay
// taken for good); threads still inside msm_amd_dist_wait return MSM_AMD_EINVAL ("the group was destroyed").  An exchange that is in
// progress is allowed to finish first -- it is bounded by the group's timeout.
int msm_amd_dist_destroy(msm_amd_ctx* ctx) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::unique_lock<std::mutex> dl(ctx->dist_mu);
    if (!ctx->dist.comm) return 0;
    ctx->dist_cv.wait(dl, [&] { return !ctx->dist.exchanging; });
    if (!ctx->dist.comm) return 0;   // (another thread destroyed it meanwhile)
    // calls whose partial is still with the engine (never waited for, or waiting for a turn that will not come): collect them into a
    // scratch buffer -- only the holder of the turn takes partials from the engine, and the turn is free here
    std::vector<uint64_t> orphans;
    for (auto& kv : ctx->dist.recs)
        if (kv.second.rc == 0 && !kv.second.collected) orphans.push_back(kv.second.engine_ticket);
    (void)hipSetDevice(ctx->device);
    if (ctx->dist.stream) {
        (void)hipStreamSynchronize(ctx->dist.stream);
        (void)hipStreamDestroy(ctx->dist.stream);
    }
    if (!ctx->dist.comm_gone) (void)rccl().CommDestroy(ctx->dist.comm);
    if (ctx->dist.d_buf) (void)hipFree(ctx->dist.d_buf);
    if (ctx->dist.h_buf) (void)hipHostFree(ctx->dist.h_buf);
    ctx->dist = msm_amd_ctx::Dist{};
    ++ctx->dist_epoch;
    dl.unlock();
    ctx->dist_cv.notify_all();
    uint8_t scratch[288];
    for (uint64_t t : orphans) (void)msm_amd_wait(ctx, t, scratch);
    return 0;
}

int msm_amd_dist_info(const msm_amd_ctx* ctx, int* rank, int* world) {
    if (!ctx || !rank || !world) return MSM_AMD_EINVAL;
    if (!dist_group(ctx, rank, world)) *rank = *world = 0;
    return 0;
}

// Ranks the communicator itself counts (ncclCommCount): what a launcher prints to show that the collective library saw all of them.
// *dead = 1 once the group has been aborted.
int msm_amd_dist_comm_count(const msm_amd_ctx* ctx, int* count, int* dead) {
    if (!ctx || !count) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> dl(ctx->dist_mu);
    *count = 0;
    if (dead) *dead = ctx->dist.dead ? 1 : 0;
    if (!ctx->dist.comm) return MSM_AMD_EINVAL;
    if (ctx->dist.comm_gone || ctx->dist.exchanging) {   // (the communicator is gone, or busy on another thread: the world it was created with)
        *count = ctx->dist.world;
        return 0;
    }
    Rccl& r = rccl();
    if (!r.CommCount || r.CommCount(ctx->dist.comm, count) != 0) *count = ctx->dist.world;
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
                uint32_t w, uint32_t gw, uint32_t gp, uint32_t n_total, uint64_t* ticket, int refused_rc = 0, const std::string& refused_why = std::string(),
                msm_amd_bases* prepared = nullptr) {
    // (dist_mu is held across the submission so that two threads cannot take sequence numbers in one order and engine tickets in
    // the other; the submission itself only queues GPU work)
    std::lock_guard<std::mutex> dl(ctx->dist_mu);
    if (!ctx->dist.comm || ctx->dist.dead) {   // (destroyed or died between the entry point's check and here)
        msm_set_err(ctx, ctx->dist.comm ? "multiexp_dist: this group is dead (" + ctx->dist.dead_why + ")" : std::string("multiexp_dist: msm_amd_dist_init first"));
        return ctx->dist.comm ? MSM_AMD_EPEER : MSM_AMD_EINVAL;
    }
    msm_amd_ctx::Dist::Rec rec;
    rec.curve = curve;
    rec.group = group;
    rec.split = gp | (gw << 16);
    rec.n_total = n_total;
    if (refused_rc) {   // the entry point has already found the call unusable on this rank
        rec.rc = refused_rc;
        rec.err = refused_why;
    } else {
        // (a prepared table holds every window of its points: such a part is a whole MSM of this rank's points)
        rec.rc = prepared ? msm_amd_multiexp_prepared_async(ctx, prepared, flags, scalars, scalar_size, count, &rec.engine_ticket)
                          : msm_amd_multiexp_shard_async(ctx, curve, group, flags, bases, scalars, scalar_size, count, w, gw, &rec.engine_ticket);
        if (rec.rc) rec.err = msm_amd_last_error(ctx);
        Geom g;
        if (!rec.rc && !prepared && msm_ticket_geom(ctx, rec.engine_ticket, &g)) rec.layout = layout_hash(curve, group, scalar_size, g);
    }
    rec.seq = ctx->dist.next_seq++;
    const uint64_t id = ctx->dist.next_id++;
    ctx->dist.recs[id] = rec;
    *ticket = id;
    return 0;
}

bool bad_kind(msm_amd_ctx* ctx, int curve, int group, uint32_t scalar_size) {
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2) || scalar_size < 1 || scalar_size > 64) {
        msm_set_err(ctx, "multiexp_dist: bad curve / group / scalar_size");
        return true;   // (the same on every rank: nothing was queued anywhere)
    }
    return false;
}

// One exchange on the group's stream, bounded: the staged 304 bytes go up, ONE all-gather, world x 304 bytes come down.  Runs OUTSIDE
// dist_mu (the caller holds the group's turn: `exchanging`).  Returns 0, or an error with `why` set; *aborted = true when the
// communicator had to be aborted (a peer that never arrived, a failure of the collective itself): it is gone then.
int exchange_once(msm_amd_ctx* ctx, void* comm, hipStream_t st, uint8_t* d, uint8_t* h, int world, uint32_t timeout_ms, std::string* why, bool* aborted) {
    *aborted = false;
    if (hipSetDevice(ctx->device) != hipSuccess) {
        *why = "dist_wait: hipSetDevice failed";
        (void)hipGetLastError();
        return MSM_AMD_EHIP;
    }
    Rccl& r = rccl();
    hipError_t e = hipMemcpyAsync(d, h, kPayload, hipMemcpyHostToDevice, st);
    int nrc = 0;
    if (e == hipSuccess) nrc = r.AllGather(d, d + kPayload, kPayload, kNcclUint8, comm, st);
    if (e == hipSuccess && nrc == 0) e = hipMemcpyAsync(h + kPayload, d + kPayload, kPayload * (size_t)world, hipMemcpyDeviceToHost, st);
    // wait for the stream, but not for ever: a peer that died, or never submitted this call, would leave this rank in the collective.
    // The exchange itself takes ~50 us: poll closely for the first milliseconds, then every 200 us.
    bool timed_out = false;
    hipError_t q = hipSuccess;
    if (e == hipSuccess && nrc == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            q = hipStreamQuery(st);
            if (q != hipErrorNotReady) break;
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (timeout_ms && ms > (double)timeout_ms) {
                timed_out = true;
                break;
            }
            if (ms > 5.0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        (void)hipGetLastError();   // (hipErrorNotReady is not an error to keep)
    }
    if (e == hipSuccess && nrc == 0 && !timed_out && q == hipSuccess) return 0;
    // something went wrong on this side or a peer did not come: abort the communicator so that whatever it queued leaves the stream,
    // then drain the stream -- after this the group is dead on this rank, and the peers find out through their own timeouts
    if (timed_out) *why = "dist_wait: the exchange did not complete within " + std::to_string(timeout_ms) + " ms (a rank of the group died, or never submitted this call; option \"dist_timeout_ms\")";
    else if (nrc != 0) *why = std::string("ncclAllGather: ") + r.GetErrorString(nrc);
    else *why = std::string("dist_wait: ") + hipGetErrorString(e != hipSuccess ? e : q);
    (void)r.CommAbort(comm);
    *aborted = true;
    (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    return timed_out ? MSM_AMD_EPEER : MSM_AMD_EHIP;
}

}  // namespace

extern "C" {

// Every rank holds ALL n points and scalars; this rank works on its slice and windows of them.
int msm_amd_multiexp_dist_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                                uint32_t n, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    int my_rank = 0, world = 0;
    if (int rc = dist_group_alive(ctx, &my_rank, &world)) return rc;
    if (bad_kind(ctx, curve, group, scalar_size)) return MSM_AMD_EINVAL;
    uint32_t gp, gw, first, count;
    msm_dist_split(ctx, curve, group, scalar_size, n, (uint32_t)world, &gp, &gw);
    msm_dist_slice(n, gp, (uint32_t)my_rank / gw, &first, &count);
    const size_t bstride = (size_t)((flags & MSM_AMD_BASES_JACOBIAN) ? 3 : 2) * (curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group;
    const uint8_t* b = bases ? (const uint8_t*)bases + (size_t)first * bstride : nullptr;
    const uint8_t* s = scalars ? (const uint8_t*)scalars + (size_t)first * scalar_size : nullptr;
    return dist_submit(ctx, curve, group, flags, b, s, scalar_size, count, (uint32_t)my_rank % gw, gw, gp, n, ticket);
}

// This rank holds only ITS slice of the n_total points (msm_amd_dist_plan says which): bases_local / scalars_local are that slice.
int msm_amd_multiexp_dist_local_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases_local, const void* scalars_local,
                                      uint32_t scalar_size, uint32_t n_local, uint32_t n_total, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    int my_rank = 0, world = 0;
    if (int rc = dist_group_alive(ctx, &my_rank, &world)) return rc;
    if (bad_kind(ctx, curve, group, scalar_size)) return MSM_AMD_EINVAL;
    uint32_t gp, gw, first, count;
    msm_dist_split(ctx, curve, group, scalar_size, n_total, (uint32_t)world, &gp, &gw);
    msm_dist_slice(n_total, gp, (uint32_t)my_rank / gw, &first, &count);
    if (n_local != count) {
        // the caller's buffers hold n_local points; reading `count` of them would run past their end (or drop points).  The split
        // depends on the context's options: plan and call must see the same ones.  Symmetric across ranks only if the caller's mistake
        // is -- so this rank still takes part in the exchange, with a failure mark.
        const std::string why = "multiexp_dist_local: n_local = " + std::to_string(n_local) + " but this rank's slice of " + std::to_string(n_total) +
                                " points is " + std::to_string(count) + " (msm_amd_dist_plan; were options changed in between?)";
        return dist_submit(ctx, curve, group, flags, nullptr, nullptr, scalar_size, count, (uint32_t)my_rank % gw, gw, gp, n_total, ticket, MSM_AMD_EINVAL, why);
    }
    return dist_submit(ctx, curve, group, flags, bases_local, scalars_local, scalar_size, count, (uint32_t)my_rank % gw, gw, gp, n_total, ticket);
}

}  // extern "C"

namespace {

// The exchange of ONE call of the group -- `rec`, the head of the sequence -- by the thread that holds the turn (`exchanging`), outside
// dist_mu: this rank's partial is taken from the engine, the headers and partials cross in one bounded all-gather, the headers are
// compared, the partials added.  Fills rec.out_rc / out_err / out; returns != 0 (with *why, *aborted) only when the GROUP has failed.
int exchange_call(msm_amd_ctx* ctx, msm_amd_ctx::Dist::Rec& rec, void* comm, hipStream_t st, uint8_t* d, uint8_t* h, int world, uint32_t timeout_ms,
                  std::string* why, bool* aborted) {
    const size_t jsz = (size_t)3 * (rec.curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)rec.group;
    uint8_t mine[kPayload] = {};
    int rc_local = rec.rc;
    std::string err_local = rec.err;
    if (rc_local == 0) {
        rc_local = msm_amd_wait(ctx, rec.engine_ticket, mine + kHeader);
        if (rc_local) err_local = msm_amd_last_error(ctx);
        rec.collected = true;
    }
    if (rc_local) {   // the point at infinity and the mark
        memset(mine, 0, sizeof mine);
        const uint32_t mark = (uint32_t)(-rc_local);
        memcpy(mine, &mark, 4);
    }
    memcpy(mine + 4, &rec.split, 4);
    memcpy(mine + 8, &rec.n_total, 4);
    memcpy(mine + 12, &rec.layout, 4);
    memcpy(h, mine, kPayload);
    if (int rc = exchange_once(ctx, comm, st, d, h, world, timeout_ms, why, aborted)) {
        rec.out_rc = rc_local ? rc_local : rc;
        rec.out_err = rc_local ? err_local : *why;
        return rc;
    }
    std::vector<uint8_t> parts((size_t)world * jsz);
    std::vector<uint32_t> hdr((size_t)world * 4);
    int failed_rank = -1;
    for (int k = 0; k < world; ++k) {
        memcpy(&hdr[(size_t)k * 4], h + kPayload * (size_t)(k + 1), 16);
        if (hdr[(size_t)k * 4] && failed_rank < 0) failed_rank = k;
        memcpy(parts.data() + (size_t)k * jsz, h + kPayload * (size_t)(k + 1) + kHeader, jsz);
    }
    // every rank compares every header with every other, so that all of them reach the same verdict
    std::string disagree;
    const uint32_t gw = std::max(1u, rec.split >> 16);
    for (int k = 0; k < world && disagree.empty(); ++k) {
        const uint32_t* a = &hdr[(size_t)k * 4];
        for (int j = k + 1; j < world && disagree.empty(); ++j) {
            const uint32_t* b = &hdr[(size_t)j * 4];
            if (a[1] != b[1] || a[2] != b[2])
                disagree = "ranks " + std::to_string(k) + " and " + std::to_string(j) + " planned different splits of this call (" + std::to_string(a[1] & 0xffff) + " x " +
                           std::to_string(a[1] >> 16) + " of " + std::to_string(a[2]) + " points against " + std::to_string(b[1] & 0xffff) + " x " + std::to_string(b[1] >> 16) +
                           " of " + std::to_string(b[2]) + "): the same options (\"dist_gw\") and arguments on every rank";
            else if ((uint32_t)k / gw == (uint32_t)j / gw && a[3] && b[3] && !a[0] && !b[0] && a[3] != b[3])
                disagree = "ranks " + std::to_string(k) + " and " + std::to_string(j) + " share a point slice but use different window layouts: the same options "
                           "(\"window_bits\", \"even_windows\") on every rank";
        }
    }
    if (rc_local) {
        rec.out_rc = rc_local;
        rec.out_err = err_local;
    } else if (failed_rank >= 0) {
        rec.out_rc = MSM_AMD_EPEER;
        rec.out_err = "multiexp_dist: rank " + std::to_string(failed_rank) + " of the group failed its part of this call (its own error text says why)";
    } else if (!disagree.empty()) {
        rec.out_rc = MSM_AMD_EINVAL;
        rec.out_err = "multiexp_dist: " + disagree;
    } else {
        rec.out.resize(jsz);
        rec.out_rc = msm_amd_combine(rec.curve, rec.group, parts.data(), (uint32_t)world, rec.out.data());
        if (rec.out_rc) rec.out_err = "multiexp_dist: combining the partials failed";
    }
    return 0;
}

}  // namespace

extern "C" {

// Takes part in the exchange of this call (one ncclAllGather of world x 304 bytes: a header and the Jacobian partial of every rank) and
// writes the normalised sum -- the same bytes on every rank.
// The exchanges of a context happen in the order in which its calls were SUBMITTED, on every rank: every rank submits the same calls
// in the same order, so the collectives match -- two promises resolved by two libuv workers cannot cross, and a G1 call cannot meet a G2
// call.  The tickets may be WAITED FOR in any order and from any threads: a thread that asks for a later ticket performs the exchanges
// of the earlier ones first and parks their results for whoever asks for them (every ticket must still be waited for, or the group
// left with msm_amd_dist_destroy).
// A rank whose own part failed still takes part, with a failure mark: it returns its error, the others MSM_AMD_EPEER.
// The exchange itself runs outside dist_mu (only the turn is held), and it is bounded: after `dist_timeout_ms` without the peers, or on
// a failure of the collective, the communicator is aborted and the group is dead -- this and every later call return an error at once.
// The headers are compared: ranks that planned different splits, or different window layouts inside one point slice, get MSM_AMD_EINVAL
// on every rank instead of a sum of partials that do not belong together.
int msm_amd_dist_wait(msm_amd_ctx* ctx, int curve, int group, uint64_t ticket, uint8_t* out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    std::unique_lock<std::mutex> dl(ctx->dist_mu);
    if (!ctx->dist.comm) {
        msm_set_err(ctx, "dist_wait: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    {
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
    }
    const uint64_t epoch = ctx->dist_epoch;
    for (;;) {
        if (ctx->dist_epoch != epoch) {   // (the records went with the group)
            msm_set_err(ctx, "dist_wait: the group was destroyed while a call was waiting for its exchange");
            return MSM_AMD_EINVAL;
        }
        auto it = ctx->dist.recs.find(ticket);   // (present: only this thread erases it)
        msm_amd_ctx::Dist::Rec& mine = it->second;
        if (mine.done) {
            const int rc = mine.out_rc;
            const std::string err = mine.out_err;
            if (rc == 0) memcpy(out, mine.out.data(), mine.out.size());
            ctx->dist.recs.erase(it);
            dl.unlock();
            if (rc) msm_set_err(ctx, err);
            return rc;
        }
        if (ctx->dist.dead) {
            // no exchange will happen any more: free this call's slot and report
            const bool collect = mine.rc == 0 && !mine.collected;
            const uint64_t et = mine.engine_ticket;
            const int rc_local = mine.rc;
            const std::string why = ctx->dist.dead_why;
            ctx->dist.recs.erase(it);
            dl.unlock();
            uint8_t scratch[288];
            if (collect) (void)msm_amd_wait(ctx, et, scratch);
            msm_set_err(ctx, "multiexp_dist: this group is dead (" + why + "): msm_amd_dist_destroy, then a new msm_amd_dist_init on every rank");
            return rc_local ? rc_local : MSM_AMD_EPEER;
        }
        if (ctx->dist.exchanging) {
            ctx->dist_cv.wait(dl);
            continue;
        }
        // the turn is free: perform the exchange at the head of the sequence -- this ticket's or an earlier one's
        uint64_t head_id = 0;
        for (auto& kv : ctx->dist.recs)
            if (!kv.second.done && kv.second.seq == ctx->dist.next_exchange) head_id = kv.first;
        if (!head_id) {   // cannot happen: this ticket itself is not done, so the sequence has not passed it
            msm_set_err(ctx, "dist_wait: internal error (no call at the head of the sequence)");
            ctx->dist.recs.erase(it);
            return MSM_AMD_EHIP;
        }
        ctx->dist.exchanging = true;
        msm_amd_ctx::Dist::Rec head = ctx->dist.recs[head_id];
        void* comm = ctx->dist.comm;
        hipStream_t st = ctx->dist.stream;
        const int world = ctx->dist.world;
        uint8_t* d = (uint8_t*)ctx->dist.d_buf;
        uint8_t* h = (uint8_t*)ctx->dist.h_buf;
        const uint32_t timeout_ms = ctx->dist_timeout_ms.load();
        dl.unlock();
        std::string why;
        bool aborted = false;
        const int grc = exchange_call(ctx, head, comm, st, d, h, world, timeout_ms, &why, &aborted);
        dl.lock();
        ctx->dist.exchanging = false;
        if (ctx->dist_epoch == epoch) {   // (a destroy waits for the turn to be handed back, so this always holds; kept as a guard)
            ++ctx->dist.next_exchange;
            head.done = true;
            auto hit = ctx->dist.recs.find(head_id);
            if (hit != ctx->dist.recs.end()) {
                head.waiting = hit->second.waiting;   // (someone may have started to wait for it meanwhile)
                hit->second = std::move(head);
            }
            if (grc) {
                ctx->dist.dead = true;
                ctx->dist.comm_gone = aborted;
                ctx->dist.dead_why = why;
            }
        }
        ctx->dist_cv.notify_all();
    }
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
    if (int rc = dist_group_alive(ctx, &my_rank, &world)) return rc;
    (void)my_rank;
    int curve = 0, group = 0;
    if (!local_table || !msm_bases_lookup(ctx, local_table, &curve, &group)) {
        msm_set_err(ctx, "multiexp_dist_prepared: not a table of this context (or already released)");
        return MSM_AMD_EINVAL;
    }
    return dist_submit(ctx, curve, group, flags, nullptr, scalars_local, scalar_size, n_local, 0, 1, (uint32_t)world, 0, ticket, 0, std::string(), local_table);
}
int msm_amd_multiexp_dist_prepared(msm_amd_ctx* ctx, msm_amd_bases* local_table, uint32_t flags, const void* scalars_local, uint32_t scalar_size,
                                   uint32_t n_local, uint8_t* out) {
    uint64_t t = 0;
    int curve = 0, group = 0;
    if (!ctx || !local_table || !msm_bases_lookup(ctx, local_table, &curve, &group)) {
        if (ctx) msm_set_err(ctx, "multiexp_dist_prepared: not a table of this context (or already released)");
        return MSM_AMD_EINVAL;
    }
    if (int rc = msm_amd_multiexp_dist_prepared_async(ctx, local_table, flags, scalars_local, scalar_size, n_local, &t)) return rc;
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

int msm_amd_multiexp_dist_local(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases_local, const void* scalars_local,
                                uint32_t scalar_size, uint32_t n_local, uint32_t n_total, uint8_t* out) {
    uint64_t t = 0;
    if (int rc = msm_amd_multiexp_dist_local_async(ctx, curve, group, flags, bases_local, scalars_local, scalar_size, n_local, n_total, &t)) return rc;
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

}  // extern "C"
