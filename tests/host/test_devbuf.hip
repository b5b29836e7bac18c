// Host-only test of the scratch-buffer bookkeeping (csrc/engine_ctx.hpp: DevBuf, ensure(), ZeroMark), no GPU involved:
// ensure() is driven through an allocator that hands the SAME address back when a buffer grows -- the sequence behind the one GPU
// memory fault of round 2 (a regrown bin histogram taken for still-zero beyond its old capacity).  "Known to be zero" must not
// survive a reallocation, whatever address it returns.
#include <cstdio>
#include <cstdlib>

#include "engine_ctx.hpp"

void msm_set_err(msm_amd_ctx* ctx, const std::string& msg) { ctx->err = msg; }

static unsigned char g_arena[1 << 20];
static int g_allocs = 0, g_frees = 0;
static hipError_t stub_alloc(void** p, size_t n) {
    if (n > sizeof g_arena) return hipErrorOutOfMemory;
    *p = g_arena;   // always the same address
    ++g_allocs;
    return hipSuccess;
}
static hipError_t stub_free(void*) {
    ++g_frees;
    return hipSuccess;
}

#define CHECK(cond)                                             \
    do {                                                        \
        if (!(cond)) {                                          \
            printf("FAILED line %d: %s\n", __LINE__, #cond);    \
            return 1;                                           \
        }                                                       \
    } while (0)

int main() {
    dev_alloc_hooks().alloc = stub_alloc;
    dev_alloc_hooks().release = stub_free;
    msm_amd_ctx* ctx = new msm_amd_ctx();
    DevBuf b;
    ZeroMark zero;
    CHECK(!zero.holds(b));                         // nothing allocated, nothing known
    CHECK(ensure(ctx, b, 33 * 4) == 0);            // the 33 bins of an 8-bit-window call
    CHECK(b.p == g_arena && b.gen == 1 && g_allocs == 1);
    CHECK(!zero.holds(b));                         // fresh memory is not known to be zero
    zero.set(b);                                   // ... k_bin_scan has left it zero
    CHECK(zero.holds(b));
    CHECK(ensure(ctx, b, 33 * 4) == 0);            // same size again: no reallocation, still known zero
    CHECK(b.gen == 1 && g_allocs == 1 && zero.holds(b));
    const size_t cap_small = b.cap;
    CHECK(ensure(ctx, b, 17 * 128 * 4) == 0);      // first 16-bit-window call: the buffer grows ...
    CHECK(g_allocs == 2 && g_frees == 1);
    CHECK(b.p == g_arena && b.cap > cap_small);    // ... and comes back at the SAME address
    CHECK(b.gen == 2);
    CHECK(!zero.holds(b));                         // the pointer matches, the generation does not: the engine will clear it
    zero.set(b);
    CHECK(zero.holds(b));
    zero.clear();                                  // a failed submission drops the mark
    CHECK(!zero.holds(b));
    CHECK(ensure(ctx, b, sizeof g_arena + 1) != 0);   // allocation failure is reported, the buffer is left empty
    CHECK(b.p == nullptr && b.cap == 0);
    zero.set(b);
    CHECK(!zero.holds(b));                         // an empty buffer is never "known zero"
    delete ctx;
    printf("devbuf: ok\n");
    return 0;
}
