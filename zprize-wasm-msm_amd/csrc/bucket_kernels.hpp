// bucket_kernels.hpp -- the two ALU-bound kernels of the engine (bucket accumulation, bucket
// reduction) and their launchers.  Each coordinate field gets its own translation unit
// (bucket_*.hip) so the heavy instantiations compile in parallel; msm_engine.hip only sees the
// launcher declarations.
#pragma once
#include "engine_common.hpp"

namespace msm {

// ------------------------------------------------------------------------------------------
// stage 4: bucket accumulation -- one lane per bucket
// ------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_accumulate(const uint8_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                     const uint32_t* __restrict__ offset, const uint32_t* __restrict__ count,
                                                     const uint32_t* __restrict__ order, uint32_t nb, uint8_t* __restrict__ buckets) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nb) return;
    uint32_t b = order[r];
    uint32_t cnt = count[b];
    uint32_t off = offset[b];
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t e = sorted[off + k];
        uint32_t idx = e & 0x7fffffffu;
        Affine<F> p = load16<Affine<F>>(bases + (size_t)idx * sizeof(Affine<F>));
        bool inf = p.is_inf();
        p.y = F::cneg(p.y, (e >> 31) != 0 && !inf);
        XYZZ<F>::madd(acc, p.x, p.y);
    }
    store16(buckets + (size_t)b * sizeof(XYZZ<F>), acc);
}

// ------------------------------------------------------------------------------------------
// stage 5: bucket reduction  S_w = sum_{k=1..half} k * B[w][k-1]
// Lane t of window w owns buckets [t*L, t*L+L): running sums give sum_j (j+1) B[t*L+j] and
// A = sum_j B[t*L+j]; adding (t*L)*A (double-and-add) makes the lane's share exact; a workgroup
// tree in LDS then leaves one partial per workgroup.
// ------------------------------------------------------------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_bucket_reduce(const uint8_t* __restrict__ buckets, uint32_t half, uint32_t L, uint32_t lanes_per_window,
                                                        uint32_t wgs_per_window, uint8_t* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    uint32_t w = blockIdx.x / wgs_per_window;
    uint32_t wg = blockIdx.x % wgs_per_window;
    uint32_t t = wg * blockDim.x + threadIdx.x;
    XYZZ<F> res = XYZZ<F>::inf();
    if (t < lanes_per_window) {
        uint32_t base = t * L;
        XYZZ<F> run = XYZZ<F>::inf();
        XYZZ<F> sum = XYZZ<F>::inf();
        for (int j = (int)L - 1; j >= 0; --j) {
            if (base + j < half) {
                XYZZ<F> bk = load16<XYZZ<F>>(buckets + ((size_t)w * half + base + j) * sizeof(XYZZ<F>));
                run = XYZZ<F>::add_ni(run, bk);
            }
            sum = XYZZ<F>::add_ni(sum, run);
        }
        // res = sum + base * run
        XYZZ<F> m = XYZZ<F>::inf();
        if (base != 0) {
            int top = 31 - __clz(base);
            for (int bit = top; bit >= 0; --bit) {
                m = XYZZ<F>::dbl_ni(m);
                if ((base >> bit) & 1u) m = XYZZ<F>::add_ni(m, run);
            }
        }
        res = XYZZ<F>::add_ni(sum, m);
    }
    sh[threadIdx.x] = res;
    __syncthreads();
    for (int d = blockDim.x / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            XYZZ<F> a = sh[threadIdx.x];
            XYZZ<F> bb = sh[threadIdx.x + d];
            sh[threadIdx.x] = XYZZ<F>::add_ni(a, bb);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) store16(partials + (size_t)blockIdx.x * sizeof(XYZZ<F>), sh[0]);
}


template <class F>
void launch_accumulate(hipStream_t st, const uint8_t* bases, const uint32_t* sorted, const uint32_t* offset, const uint32_t* count,
                       const uint32_t* order, uint32_t nb, uint8_t* buckets) {
    k_accumulate<F><<<(nb + 255) / 256, 256, 0, st>>>(bases, sorted, offset, count, order, nb, buckets);
}
template <class F>
void launch_bucket_reduce(hipStream_t st, uint32_t nparts, uint32_t rblock, const uint8_t* buckets, uint32_t half, uint32_t L,
                          uint32_t lanes_per_window, uint32_t wgs_per_window, uint8_t* partials) {
    k_bucket_reduce<F><<<nparts, rblock, (size_t)rblock * sizeof(XYZZ<F>), st>>>(buckets, half, L, lanes_per_window, wgs_per_window, partials);
}

#define MSM_INSTANTIATE_BUCKET_KERNELS(F)                                                                                                   \
    template void launch_accumulate<F>(hipStream_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint32_t*, const uint32_t*,   \
                                       uint32_t, uint8_t*);                                                                                 \
    template void launch_bucket_reduce<F>(hipStream_t, uint32_t, uint32_t, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t*);

}  // namespace msm
