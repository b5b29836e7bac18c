// Bucket kernels for BN254 G1 (254-bit Fp, multiplies expanded inline).
#include "bucket_kernels.hpp"
namespace msm {
using F_ = Fp<Bn254Fp, true>;
MSM_INSTANTIATE_ACCUMULATE(F_, 0)        // two-chain variant for launches of one wave per SIMD
MSM_INSTANTIATE_SLICES(F_, 0)            // several lanes per bucket for launches with few buckets (lanes short of two waves per SIMD)
MSM_INSTANTIATE_JOIN(F_)
MSM_INSTANTIATE_ITEMS(F_, 0)           // the work list: lanes per bucket in proportion to its population (two-chain build)
MSM_INSTANTIATE_JOIN_ITEMS(F_)
MSM_INSTANTIATE_REDUCE_PAIR(F_)            // the reduction with every point on a lane pair
MSM_INSTANTIATE_TAIL_AND_REDUCE(F_)   // the accumulation kernel itself: bucket_g1_bn254_acc.hip
}  // namespace msm
