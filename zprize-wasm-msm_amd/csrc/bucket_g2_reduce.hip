// Bucket REDUCTION kernels for G2 of both curves (Fp2 coordinates on lane pairs): additions and base-field products expanded inline
// in the kernel, as for G1 (a kernel may be large; it is device FUNCTIONS beyond 128 KB that hang this GPU).
#define MSM_G2_REDUCE_INLINE 1
#include "bucket_kernels.hpp"
namespace msm {
using F2a_ = Fp2<Bls12381Fp>;
using F2b_ = Fp2<Bn254Fp>;
MSM_INSTANTIATE_REDUCE(F2a_)
MSM_INSTANTIATE_REDUCE(F2b_)
MSM_INSTANTIATE_JOIN_ITEMS(F2a_)   // the join of the work list: one inlined addition per lane, no private segment
MSM_INSTANTIATE_JOIN_ITEMS(F2b_)
}  // namespace msm
