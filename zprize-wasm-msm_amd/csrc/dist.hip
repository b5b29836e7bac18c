// dist.hip -- the multi-GPU exchange behind the C ABI (the seam is the reference's per-window entry point
// fnName_chunk, wasmcurves/src/build_multiexp.js:96-104, which its external caller fans out to workers and recombines).
//
// One process per GPU.  Rank r of `world` computes the windows {w : w mod world = r} of the same (bases, scalars) on its own
// MI355X (msm_amd_multiexp_shard_async), which leaves ONE Jacobian partial, already weighted by 2^(c*w).  The partials -- 144 bytes
// for BLS12-381 G1, at most 288 for G2 -- cross xGMI in ONE ncclAllGather (elliptic-curve addition is not an RCCL reduction
// operator, hence gather + add), and every rank adds them on the host.  Latency-bound by construction: world x 144 bytes.
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

int nccl_fail(msm_amd_ctx* ctx, const char* what, int rc) {
    msm_set_err(ctx, std::string(what) + ": " + rccl().GetErrorString(rc));
    return MSM_AMD_EHIP;
}

}  // namespace

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
    std::lock_guard<std::mutex> lk(ctx->mu);
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
    msm_nccl_id id;
    memcpy(id.internal, id128, sizeof id.internal);
    void* comm = nullptr;
    int rc = r.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return nccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->dist.comm = comm;
    ctx->dist.rank = rank;
    ctx->dist.world = world;
    const size_t cap = (size_t)(world + 1) * 288;   // the largest partial is a BLS12-381 G2 Jacobian point
    hipError_t e = hipMalloc(&ctx->dist.d_buf, cap * msm_amd_ctx::MAX_SLOTS);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->dist.h_buf, cap * msm_amd_ctx::MAX_SLOTS, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->dist.stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        msm_set_err(ctx, std::string("dist_init: ") + hipGetErrorString(e));
        return MSM_AMD_EHIP;
    }
    ctx->dist.slot_bytes = cap;
    return 0;
}

int msm_amd_dist_destroy(msm_amd_ctx* ctx) {
    if (!ctx) return MSM_AMD_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
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
    return 0;
}

int msm_amd_dist_info(const msm_amd_ctx* ctx, int* rank, int* world) {
    if (!ctx || !rank || !world) return MSM_AMD_EINVAL;
    *rank = ctx->dist.comm ? ctx->dist.rank : 0;
    *world = ctx->dist.comm ? ctx->dist.world : 0;
    return 0;
}

// This rank's windows on this rank's GPU; the ticket is collected with msm_amd_dist_wait.
int msm_amd_multiexp_dist_async(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                                uint32_t n, uint64_t* ticket) {
    if (!ctx || !ticket) return MSM_AMD_EINVAL;
    if (!ctx->dist.comm) {
        msm_set_err(ctx, "multiexp_dist: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    return msm_amd_multiexp_shard_async(ctx, curve, group, flags, bases, scalars, scalar_size, n, (uint32_t)ctx->dist.rank, (uint32_t)ctx->dist.world,
                                        ticket);
}

// Waits for this rank's partial, exchanges the partials of all ranks (one ncclAllGather of world x jacobian bytes) and writes the
// normalised sum -- the same bytes on every rank.  Every rank must wait for its tickets in the same order.
int msm_amd_dist_wait(msm_amd_ctx* ctx, int curve, int group, uint64_t ticket, uint8_t* out) {
    if (!ctx || !out) return MSM_AMD_EINVAL;
    if (!ctx->dist.comm) {
        msm_set_err(ctx, "dist_wait: msm_amd_dist_init first");
        return MSM_AMD_EINVAL;
    }
    if ((curve != MSM_AMD_BLS12381 && curve != MSM_AMD_BN128) || (group != 1 && group != 2)) {
        msm_set_err(ctx, "unknown curve/group");
        return MSM_AMD_EINVAL;
    }
    const size_t jsz = (size_t)3 * (curve == MSM_AMD_BLS12381 ? 48 : 32) * (size_t)group;
    uint8_t mine[288];
    if (int rc = msm_amd_wait(ctx, ticket, mine)) return rc;
    // the exchanges of one context are serialised (collectives of one communicator must be issued in the same order on every rank)
    std::lock_guard<std::mutex> lk(ctx->dist_mu);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int world = ctx->dist.world;
    uint8_t* d = (uint8_t*)ctx->dist.d_buf;
    uint8_t* h = (uint8_t*)ctx->dist.h_buf;
    memcpy(h, mine, jsz);
    HIPCHK(ctx, hipMemcpyAsync(d, h, jsz, hipMemcpyHostToDevice, ctx->dist.stream));
    int rc = rccl().AllGather(d, d + 288, jsz, kNcclUint8, ctx->dist.comm, ctx->dist.stream);
    if (rc != 0) return nccl_fail(ctx, "ncclAllGather", rc);
    HIPCHK(ctx, hipMemcpyAsync(h + 288, d + 288, jsz * (size_t)world, hipMemcpyDeviceToHost, ctx->dist.stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->dist.stream));
    return msm_amd_combine(curve, group, h + 288, (uint32_t)world, out);
}

int msm_amd_multiexp_dist(msm_amd_ctx* ctx, int curve, int group, uint32_t flags, const void* bases, const void* scalars, uint32_t scalar_size,
                          uint32_t n, uint8_t* out) {
    uint64_t t = 0;
    if (int rc = msm_amd_multiexp_dist_async(ctx, curve, group, flags, bases, scalars, scalar_size, n, &t)) return rc;
    return msm_amd_dist_wait(ctx, curve, group, t, out);
}

}  // extern "C"
