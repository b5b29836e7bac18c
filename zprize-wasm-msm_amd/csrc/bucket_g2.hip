// Tail kernels for G2 of both curves (Fp2 coordinates on lane pairs, curve_u.hpp UG2P).  These kernels run
// few waves and chain full additions through out-of-line functions; the base-field products stay out of line too, which
// keeps every device function far below the 128 KB that have hung this GPU.
#define MSM_U_OUTLINE_PRODUCTS 1
#include "bucket_kernels.hpp"
namespace msm {
using F2a_ = Fp2<Bls12381Fp>;
using F2b_ = Fp2<Bn254Fp>;
MSM_INSTANTIATE_TAIL(F2a_)   // the accumulation kernels: bucket_g2_acc.hip; the reduction: bucket_g2_reduce.hip
MSM_INSTANTIATE_TAIL(F2b_)
MSM_INSTANTIATE_JOIN(F2a_)   // the slices of a bucket joined (out-of-line additions, like the tail kernels)
MSM_INSTANTIATE_JOIN(F2b_)
}  // namespace msm
