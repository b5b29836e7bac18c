// Tail and join kernels for BLS12-381 G2 (Fp2 coordinates on lane pairs, curve_u.hpp UG2P): additions AND base-field products expanded
// inline in the kernels, like the reduction (bucket_g2_reduce.hip) -- no out-of-line call, hence no private segment and no per-queue
// scratch (bucket_kernels.hpp, tadd).  A kernel may be large; it is device FUNCTIONS beyond 128 KB that have hung this GPU.
#include "bucket_kernels.hpp"
namespace msm {
using F2a_ = Fp2<Bls12381Fp>;
MSM_INSTANTIATE_TAIL(F2a_)   // the accumulation kernels: bucket_g2_acc.hip; the reduction and the work list's join: bucket_g2_reduce.hip
MSM_INSTANTIATE_JOIN(F2a_)   // the slices of a bucket joined
}  // namespace msm
