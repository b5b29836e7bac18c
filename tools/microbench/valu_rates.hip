// Microbenchmark: issue rate of the integer / fp64 VALU instructions the Fp Montgomery
// multiplier can be built from, on gfx950.  Not part of the product; informs DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc0 = a, acc1 = b, acc2 = a ^ b, acc3 = a + b, acc4 = 5, acc5 = 6, acc6 = 7, acc7 = 8;
    uint32_t r0 = a, r1 = b, r2 = a ^ 1, r3 = b ^ 2, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5;
    double da = 1.0000001, db = 0.5;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_mad_u64_u32
            REP8(
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc2) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc3) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc4) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc5) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc6) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc7) : "v"(a), "v"(b) : "vcc");
            )
        } else if (OP == 1) {  // v_mul_lo_u32
            REP8(
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r0) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r1) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r2) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r3) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r4) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r5) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r6) : "v"(a));
            asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r7) : "v"(a));
            )
        } else if (OP == 2) {  // v_mul_hi_u32
            REP8(
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r0) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r1) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r2) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r3) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r4) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r5) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r6) : "v"(a));
            asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r7) : "v"(a));
            )
        } else if (OP == 3) {  // v_mad_u32_u24
            REP8(
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r4) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r5) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r6) : "v"(a), "v"(b));
            asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r7) : "v"(a), "v"(b));
            )
        } else if (OP == 4) {  // v_fma_f64
            REP8(
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d4) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d5) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d6) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d7) : "v"(da), "v"(db));
            )
        } else if (OP == 5) {  // v_add_co_u32 + v_addc_co_u32 pair (counts as 2 ops)
            REP8(
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r0), "+v"(r1) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r4), "+v"(r5) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
            )
        } else if (OP == 6) {  // v_add3_u32
            REP8(
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r1) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r2) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r3) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r4) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r5) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r6) : "v"(a), "v"(b));
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r7) : "v"(a), "v"(b));
            )
        } else if (OP == 7) {  // v_lshl_add_u64 (64-bit add)
            REP8(
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc0) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc1) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc2) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc3) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc4) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc5) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc6) : "v"(acc7));
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc6) : "v"(acc7));
            )
        } else if (OP == 8) {  // v_mul_u32_u24
            REP8(
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r0) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r1) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r2) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r3) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r4) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r5) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r6) : "v"(a));
            asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r7) : "v"(a));
            )
        } else if (OP == 9) {  // v_mul_hi_u32_u24
            REP8(
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r0) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r1) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r2) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r3) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r4) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r5) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r6) : "v"(a));
            asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r7) : "v"(a));
            )
        } else if (OP == 10) {  // mixed: 1 mad_u64_u32 + 2 adds (the CIOS inner step)
            REP8(
            asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_add_co_u32 %1, vcc, %1, %3\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc" : "+v"(acc0), "+v"(r0), "+v"(r1) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_add_co_u32 %1, vcc, %1, %3\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc" : "+v"(acc1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_add_co_u32 %1, vcc, %1, %3\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc" : "+v"(acc2), "+v"(r4), "+v"(r5) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_add_co_u32 %1, vcc, %1, %3\n\tv_addc_co_u32 %2, vcc, %2, %4, vcc" : "+v"(acc3), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
            )
        }
    }
    uint64_t s = acc0 ^ acc1 ^ acc2 ^ acc3 ^ acc4 ^ acc5 ^ acc6 ^ acc7;
    uint32_t r = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    double d = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ r ^ (uint32_t)d;
}

template <int OP>
int run(const char* name, int ops_per_iter, int waves_per_simd) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave/SIMD per block
    int iters = 2000;
    uint32_t* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<OP><<<blocks, 256>>>(out, 10, 1); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(e0));
        k<OP><<<blocks, 256>>>(out, iters, 1 + r);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double ops = (double)blocks * 256 * iters * ops_per_iter;
    double rate = ops / (best * 1e-3);
    // cycles per wave-instruction per SIMD assuming the clock is 2.4 GHz
    double wave_instr = ops / 64.0;
    double simds = cus * 4.0;
    double cyc = (best * 1e-3) * 2.4e9 / (wave_instr / simds);
    printf("%-28s waves/SIMD=%d  %8.3f ms  %8.2f Tops/s  %6.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name, waves_per_simd, best, rate / 1e12, cyc);
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    for (int w : {1, 2, 4, 8}) {
        if (w == 1) {
            run<0>("v_mad_u64_u32", 64, 1); run<1>("v_mul_lo_u32", 64, 1); run<2>("v_mul_hi_u32", 64, 1);
            run<3>("v_mad_u32_u24", 64, 1); run<4>("v_fma_f64", 64, 1); run<5>("v_add_co+addc (2 ops)", 64, 1);
            run<6>("v_add3_u32", 64, 1); run<7>("v_lshl_add_u64", 64, 1); run<8>("v_mul_u32_u24", 64, 1);
            run<9>("v_mul_hi_u32_u24", 64, 1); run<10>("mad64+add+addc (3 ops)", 96, 1);
        } else if (w == 2) {
            run<0>("v_mad_u64_u32", 64, 2); run<4>("v_fma_f64", 64, 2); run<6>("v_add3_u32", 64, 2); run<10>("mad64+add+addc (3 ops)", 96, 2);
        } else if (w == 4) {
            run<0>("v_mad_u64_u32", 64, 4); run<1>("v_mul_lo_u32", 64, 4); run<2>("v_mul_hi_u32", 64, 4);
            run<3>("v_mad_u32_u24", 64, 4); run<4>("v_fma_f64", 64, 4); run<5>("v_add_co+addc (2 ops)", 64, 4);
            run<6>("v_add3_u32", 64, 4); run<7>("v_lshl_add_u64", 64, 4); run<8>("v_mul_u32_u24", 64, 4);
            run<9>("v_mul_hi_u32_u24", 64, 4); run<10>("mad64+add+addc (3 ops)", 96, 4);
        } else {
            run<0>("v_mad_u64_u32", 64, 8); run<4>("v_fma_f64", 64, 8); run<6>("v_add3_u32", 64, 8); run<10>("mad64+add+addc (3 ops)", 96, 8);
        }
    }
    return 0;
}
