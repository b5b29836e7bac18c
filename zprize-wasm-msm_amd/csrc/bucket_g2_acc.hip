// Bucket ACCUMULATION kernels for G2 of both curves: Fp2 coordinates distributed over lane pairs (field_u.hpp Fp2d), everything
// inline.
#include "bucket_kernels.hpp"
namespace msm {
using F2a_ = Fp2<Bls12381Fp>;
using F2b_ = Fp2<Bn254Fp>;
MSM_INSTANTIATE_ACCUMULATE(F2a_, 0)
MSM_INSTANTIATE_ACCUMULATE(F2b_, 0)
MSM_INSTANTIATE_SLICES(F2a_, 0)   // buckets cut into slices for launches with few buckets (small n in 8-bit windows, sharded ranks)
MSM_INSTANTIATE_SLICES(F2b_, 0)
MSM_INSTANTIATE_ITEMS(F2a_, 0)    // the work list: lanes per bucket in proportion to its population
MSM_INSTANTIATE_ITEMS(F2b_, 0)
}  // namespace msm
