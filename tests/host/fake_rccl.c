/* fake_rccl.c -- TEST INFRASTRUCTURE, never part of the product path.
 *
 * A stand-in for the seven RCCL entry points csrc/dist.hip resolves with dlsym (ncclGetUniqueId, ncclCommInitRank, ncclAllGather,
 * ncclCommDestroy, ncclCommAbort, ncclCommCount, ncclGetErrorString), carried over a POSIX shared-memory segment instead of xGMI,
 * so that the multi-rank code of dist.hip runs with two (or more) REAL processes on a box with ONE GPU -- RCCL itself refuses two
 * ranks on one device.  Selected with MSM_AMD_RCCL_LIB=<path to libfake_rccl.so>; tests/test_gpu_two_ranks.py builds and uses it.
 *
 * Semantics kept from the real library, because dist.hip depends on them:
 *   - ncclAllGather is ASYNCHRONOUS and stream-ordered: it queues a device-to-host copy, a host function (hipLaunchHostFunc) that
 *     performs the rendezvous through shared memory, and a host-to-device copy on the caller's stream, and returns at once;
 *   - a rank whose peers never arrive stays in the collective (the host function spins) until ncclCommAbort -- which, as in RCCL, is
 *     LOCAL: the peers only find out through their own timeouts;
 *   - collectives of one communicator pair up in issue order on every rank.
 * FAKE_RCCL_TIMEOUT_MS (default 120000) bounds every spin as a last resort, so that a broken test cannot hang a GPU box.
 *
 * Build: gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt -lpthread
 */
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

enum { FAKE_MAX_RANKS = 16, FAKE_MAX_BYTES = 1024, FAKE_RING = 8 };
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6 };

typedef struct {
    char internal[128];
} ncclUniqueId;

typedef struct {
    _Atomic uint32_t magic;
    _Atomic uint32_t joined;                                  /* ranks that have called ncclCommInitRank */
    _Atomic uint32_t left;                                    /* ranks that have destroyed / aborted their communicator */
    _Atomic uint64_t ready[FAKE_RING][FAKE_MAX_RANKS];        /* ready[slot][r] = 1 + the last sequence number rank r has published in that slot */
    uint8_t data[FAKE_RING][FAKE_MAX_RANKS][FAKE_MAX_BYTES];
} Shm;

typedef struct Comm {
    Shm* shm;
    char name[128];
    int rank, nranks;
    uint64_t seq;               /* collectives issued by this rank */
    _Atomic int abort;          /* ncclCommAbort: every spin of this communicator gives up */
    _Atomic int async_err;
    uint8_t* h_send;            /* pinned: [FAKE_RING][FAKE_MAX_BYTES] */
    uint8_t* h_recv;            /* pinned: [FAKE_RING][FAKE_MAX_RANKS * FAKE_MAX_BYTES] */
    uint64_t timeout_ms;
} Comm;

typedef struct {
    Comm* c;
    uint64_t seq;
    size_t bytes;
} Call;

static int dbg(void) {
    static int v = -1;
    if (v < 0) v = getenv("FAKE_RCCL_DEBUG") ? 1 : 0;
    return v;
}
#define DBG(...)                                   \
    do {                                           \
        if (dbg()) {                               \
            fprintf(stderr, "fake_rccl[%d]: ", (int)getpid()); \
            fprintf(stderr, __VA_ARGS__);          \
            fputc('\n', stderr);                   \
            fflush(stderr);                        \
        }                                          \
    } while (0)

static uint64_t now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000u + (uint64_t)ts.tv_nsec / 1000000u;
}
static void nap(void) {
    struct timespec ts = {0, 20000};   /* 20 us */
    nanosleep(&ts, NULL);
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case ncclSuccess: return "no error (fake_rccl)";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "unhandled system error (fake_rccl)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        case ncclRemoteError: return "remote error: a peer did not arrive (fake_rccl)";
        default: return "internal error (fake_rccl)";
    }
}

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "/msm_fake_rccl_%d_%lx", (int)getpid(), (unsigned long)ts.tv_nsec ^ (unsigned long)ts.tv_sec << 20);
    int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t)sizeof(Shm)) != 0) {
        close(fd);
        shm_unlink(id->internal);
        return ncclSystemError;
    }
    close(fd);   /* (a fresh segment is all zero) */
    return ncclSuccess;
}

int ncclCommInitRank(void** comm_out, int nranks, ncclUniqueId id, int rank) {
    if (!comm_out || nranks < 1 || nranks > FAKE_MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    Shm* shm = (Shm*)mmap(NULL, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (shm == MAP_FAILED) return ncclSystemError;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    if (!c) return ncclSystemError;
    c->shm = shm;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->rank = rank;
    c->nranks = nranks;
    const char* t = getenv("FAKE_RCCL_TIMEOUT_MS");
    c->timeout_ms = t ? strtoull(t, NULL, 10) : 120000u;
    if (hipHostMalloc((void**)&c->h_send, (size_t)FAKE_RING * FAKE_MAX_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_recv, (size_t)FAKE_RING * FAKE_MAX_RANKS * FAKE_MAX_BYTES, hipHostMallocDefault) != hipSuccess) {
        if (c->h_send) (void)hipHostFree(c->h_send);
        munmap(shm, sizeof(Shm));
        free(c);
        return ncclUnhandledCudaError;
    }
    atomic_fetch_add(&shm->joined, 1u);
    /* collective, like the real call: returns when every rank has joined */
    const uint64_t t0 = now_ms();
    while (atomic_load(&shm->joined) < (uint32_t)nranks) {
        if (now_ms() - t0 > c->timeout_ms) {
            (void)hipHostFree(c->h_send);
            (void)hipHostFree(c->h_recv);
            munmap(shm, sizeof(Shm));
            free(c);
            return ncclRemoteError;
        }
        nap();
    }
    *comm_out = c;
    DBG("rank %d of %d joined %s", rank, nranks, c->name);
    return ncclSuccess;
}

/* the rendezvous of one collective, on HIP's callback thread: publish this rank's bytes, wait for everybody's, gather them */
static void rendezvous(void* arg) {
    Call* k = (Call*)arg;
    Comm* c = k->c;
    Shm* shm = c->shm;
    const int slot = (int)(k->seq % FAKE_RING);
    DBG("rank %d: rendezvous %llu begins", c->rank, (unsigned long long)k->seq);
    memcpy(shm->data[slot][c->rank], c->h_send + (size_t)slot * FAKE_MAX_BYTES, k->bytes);
    atomic_store_explicit(&shm->ready[slot][c->rank], k->seq + 1, memory_order_release);
    const uint64_t t0 = now_ms();
    uint8_t* dst = c->h_recv + (size_t)slot * FAKE_MAX_RANKS * FAKE_MAX_BYTES;
    for (int r = 0; r < c->nranks; ++r) {
        while (atomic_load_explicit(&shm->ready[slot][r], memory_order_acquire) < k->seq + 1) {
            if (atomic_load(&c->abort) || now_ms() - t0 > c->timeout_ms) {
                DBG("rank %d: rendezvous %llu gives up waiting for rank %d (abort %d)", c->rank, (unsigned long long)k->seq, r, atomic_load(&c->abort));
                atomic_store(&c->async_err, ncclRemoteError);
                memset(dst, 0xEE, (size_t)c->nranks * k->bytes);   /* nobody may mistake this for data */
                free(k);
                return;
            }
            nap();
        }
        memcpy(dst + (size_t)r * k->bytes, shm->data[slot][r], k->bytes);
    }
    DBG("rank %d: rendezvous %llu complete", c->rank, (unsigned long long)k->seq);
    free(k);
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || !send || !recv) return ncclInvalidArgument;
    if (dtype != 0 && dtype != 1) return ncclInvalidArgument;   /* ncclInt8 / ncclUint8: bytes */
    if (count == 0 || count > FAKE_MAX_BYTES) return ncclInvalidArgument;
    if (atomic_load(&c->abort)) return ncclInvalidUsage;
    Call* k = (Call*)malloc(sizeof(Call));
    if (!k) return ncclSystemError;
    k->c = c;
    k->seq = c->seq++;
    k->bytes = count;
    DBG("rank %d: all-gather %llu of %zu bytes queued", c->rank, (unsigned long long)k->seq, count);
    const int slot = (int)(k->seq % FAKE_RING);
    if (hipMemcpyAsync(c->h_send + (size_t)slot * FAKE_MAX_BYTES, send, count, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipLaunchHostFunc(st, rendezvous, k) != hipSuccess) {
        free(k);
        return ncclUnhandledCudaError;
    }
    if (hipMemcpyAsync(recv, c->h_recv + (size_t)slot * FAKE_MAX_RANKS * FAKE_MAX_BYTES, count * (size_t)c->nranks, hipMemcpyHostToDevice, st) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

int ncclCommCount(const void* comm, int* count) {
    const Comm* c = (const Comm*)comm;
    if (!c || !count) return ncclInvalidArgument;
    *count = (int)atomic_load(&c->shm->joined);
    return ncclSuccess;
}

static int leave(Comm* c) {
    /* the last rank out removes the segment's name (the mapping of the others stays valid until they unmap it) */
    if (atomic_fetch_add(&c->shm->left, 1u) + 1u >= (uint32_t)c->nranks) shm_unlink(c->name);
    munmap(c->shm, sizeof(Shm));
    (void)hipHostFree(c->h_send);
    (void)hipHostFree(c->h_recv);
    free(c);
    return ncclSuccess;
}

/* The caller synchronises its stream before this, as dist.hip does (nothing of this communicator may still be queued). */
int ncclCommDestroy(void* comm) {
    if (!comm) return ncclInvalidArgument;
    return leave((Comm*)comm);
}

/* Gives up every collective of THIS rank that is waiting for its peers and frees the communicator.  The callbacks still queued on a
 * stream need the structure until they have run, so the caller's hipStreamSynchronize comes first in the real library's contract too;
 * here the flag is raised, the stream owner drains, and the memory is released a little later by whoever destroys the process -- a
 * handful of bytes per aborted communicator, in a test. */
int ncclCommAbort(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclInvalidArgument;
    atomic_store(&c->abort, 1);
    if (atomic_fetch_add(&c->shm->left, 1u) + 1u >= (uint32_t)c->nranks) shm_unlink(c->name);
    return ncclSuccess;   /* (deliberately not freed: see above) */
}
