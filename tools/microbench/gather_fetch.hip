// Microbenchmark: what FETCH_SIZE (rocprofv3 --pmc) reports for a 16-byte-per-lane GATHER with a known byte count, on gfx950.
// Not part of the product; calibrates the `traffic` figure of the bucket accumulation.
//
// The guide's correction (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read --
// double it") was established on streaming reads.  The accumulation gathers 96-byte records with 16-byte loads at random addresses.
// Here every lane reads whole records of REC bytes at random record indices, each record exactly ONCE per launch, from a table far
// larger than the 256-MiB Infinity Cache, so the bytes that must come from memory are known:
//   k_gather<128>: 128-byte records, 128-byte aligned -> every 128-byte line is needed exactly once: true bytes = table size
//   k_gather<96>:  96-byte records back to back (the accumulation's layout): 3 of 4 lines are shared by two records
//   k_stream:      the same table read front to back with 16-byte loads (the case the guide calibrated)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and compare the counter with the sizes this program prints.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                            \
    do {                                                                    \
        hipError_t e = (x);                                                 \
        if (e != hipSuccess) {                                              \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

template <int REC>
__global__ void __launch_bounds__(256) k_gather(const uint8_t* __restrict__ table, const uint32_t* __restrict__ perm, uint32_t n, uint32_t per_lane,
                                                 uint4* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t i = t * per_lane + k;
        if (i >= n) break;
        const uint4* p = reinterpret_cast<const uint4*>(table + (size_t)perm[i] * REC);
#pragma unroll
        for (int q = 0; q < REC / 16; ++q) {
            uint4 v = p[q];
            acc.x ^= v.x;
            acc.y += v.y;
            acc.z ^= v.z;
            acc.w += v.w;
        }
    }
    out[t] = acc;
}
__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, size_t n16, uint4* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = t; i < n16; i += stride) {
        uint4 v = table[i];
        acc.x ^= v.x;
        acc.y += v.y;
        acc.z ^= v.z;
        acc.w += v.w;
    }
    out[t] = acc;
}

int main() {
    const uint32_t n = 1u << 24;   // records
    const size_t bytes = (size_t)n * 128;   // 2 GiB table (used as n x 128 B or as (n * 128 / 96) x 96 B)
    uint8_t* table;
    uint32_t *perm128, *perm96;
    uint4* out;
    const uint32_t n96 = (uint32_t)(bytes / 96);
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 1, bytes));
    CHECK(hipMalloc(&perm128, (size_t)n * 4));
    CHECK(hipMalloc(&perm96, (size_t)n96 * 4));
    const uint32_t per_lane = 16;
    // one uint4 per lane of the LARGEST launch (k_gather<96>: n96 / per_lane lanes, rounded up to whole workgroups)
    const size_t max_lanes = ((size_t)(n96 / per_lane + 255) / 256 + 1) * 256;
    if (max_lanes < (size_t)4096 * 256 || max_lanes < (size_t)(n / per_lane + 255) / 256 * 256) return 1;
    CHECK(hipMalloc(&out, max_lanes * 16));
    // random permutations (each record read exactly once): i -> (a * i + b) mod m with a coprime to m, then a xorshift scramble kept inside
    // the range by cycle walking would be overkill here -- an affine map with a large odd multiplier already destroys all locality
    auto fill = [&](uint32_t* d, uint32_t m) {
        std::vector<uint32_t> h(m);
        uint64_t a = 2654435761ull % m;
        while (true) {   // make a coprime to m
            uint64_t x = a, y = m;
            while (y) { uint64_t tt = x % y; x = y; y = tt; }
            if (x == 1) break;
            ++a;
        }
        for (uint32_t i = 0; i < m; ++i) h[i] = (uint32_t)((a * i + 12345u) % m);
        return hipMemcpy(d, h.data(), (size_t)m * 4, hipMemcpyHostToDevice);
    };
    CHECK(fill(perm128, n));
    CHECK(fill(perm96, n96));
    for (int rep = 0; rep < 3; ++rep) {
        k_gather<128><<<(n / per_lane + 255) / 256, 256>>>(table, perm128, n, per_lane, out);
        k_gather<96><<<(n96 / per_lane + 255) / 256, 256>>>(table, perm96, n96, per_lane, out);
        k_stream<<<4096, 256>>>((const uint4*)table, bytes / 16, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("table %.1f MiB; per launch: k_gather<128> needs %.1f MiB (+ %.1f MiB of indices), k_gather<96> %.1f MiB (+ %.1f), k_stream %.1f MiB\n",
           bytes / 1048576.0, bytes / 1048576.0, n * 4 / 1048576.0, (double)n96 * 96 / 1048576.0, n96 * 4 / 1048576.0, bytes / 1048576.0);
    return 0;
}
