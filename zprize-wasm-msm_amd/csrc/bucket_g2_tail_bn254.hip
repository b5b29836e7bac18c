// Tail and join kernels for BN254 G2 (Fp2 coordinates on lane pairs): everything inline, see bucket_g2_tail_bls12381.hip.
#include "bucket_kernels.hpp"
namespace msm {
using F2b_ = Fp2<Bn254Fp>;
MSM_INSTANTIATE_TAIL(F2b_)
MSM_INSTANTIATE_JOIN(F2b_)
}  // namespace msm
