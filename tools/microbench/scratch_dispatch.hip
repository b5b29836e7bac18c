// scratch_dispatch.hip -- what a kernel's private segment costs at DISPATCH on gfx950 (round 4).
// The tail / join kernels of the engine call out-of-line point additions and so carry 450-1030 bytes of private segment per lane.  The
// HSA runtime sizes a queue's scratch for the whole device (bytes per lane x 64 x wave slots of the chip); a request beyond its
// single-allocation limit is served "use once": allocated for the dispatch, handed back after it.  This program times back-to-back and
// one-at-a-time launches of a tiny grid of a kernel with P bytes of private segment per lane, P = 0 ... 2048, and prints the free device
// memory before / during / after, so that the cost and the retention of that scratch are numbers rather than guesses.
// build: hipcc -O2 --offload-arch=gfx950 scratch_dispatch.hip -o scratch_dispatch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

template <int WORDS>
__global__ void k_priv(unsigned* out, unsigned n, unsigned sel) {
    if (n == 0) return;   // (the launches below pass n = 0: the waves leave at once -- only the dispatch is measured)
    volatile unsigned buf[WORDS > 0 ? WORDS : 1];
    for (int i = 0; i < (WORDS > 0 ? WORDS : 1); ++i) buf[i] = i * sel + threadIdx.x;
    unsigned acc = 0;
    for (unsigned i = 0; i < n; ++i) acc += buf[(i * sel + acc) % (WORDS > 0 ? WORDS : 1)];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static size_t free_mb() {
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    return f >> 20;
}

template <int WORDS>
static void run(hipStream_t st, unsigned* out, unsigned grid, unsigned block) {
    const size_t f0 = free_mb();
    k_priv<WORDS><<<grid, block, 0, st>>>(out, 0, 1);
    (void)hipStreamSynchronize(st);
    const size_t f1 = free_mb();
    const int reps = 200;
    double t0 = now_us();
    for (int i = 0; i < reps; ++i) k_priv<WORDS><<<grid, block, 0, st>>>(out, 0, 1);
    (void)hipStreamSynchronize(st);
    const double burst = (now_us() - t0) / reps;
    t0 = now_us();
    for (int i = 0; i < reps; ++i) {
        k_priv<WORDS><<<grid, block, 0, st>>>(out, 0, 1);
        (void)hipStreamSynchronize(st);
    }
    const double lone = (now_us() - t0) / reps;
    // alternating with a kernel that needs none: does the scratch come and go?
    t0 = now_us();
    for (int i = 0; i < reps; ++i) {
        k_priv<WORDS><<<grid, block, 0, st>>>(out, 0, 1);
        k_priv<0><<<grid, block, 0, st>>>(out, 0, 1);
    }
    (void)hipStreamSynchronize(st);
    const double alt = (now_us() - t0) / reps;
    printf("private %5d B/lane, grid %5u x %4u: back-to-back %7.2f us/launch, one at a time %7.2f us, alternating pair %7.2f us; free MB %zu -> %zu -> %zu\n", WORDS * 4, grid, block,
           burst, lone, alt, f0, f1, free_mb());
}

int main() {
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* out;
    (void)hipMalloc(&out, 1 << 26);
    for (unsigned grid : {4u, 64u, 1024u, 16384u}) {
        run<0>(st, out, grid, 256);
        run<12>(st, out, grid, 256);
        run<64>(st, out, grid, 256);
        run<128>(st, out, grid, 256);
        run<256>(st, out, grid, 256);
        run<512>(st, out, grid, 256);
    }
    return 0;
}
