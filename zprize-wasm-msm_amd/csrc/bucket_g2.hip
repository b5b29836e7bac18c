// Bucket kernels for G2 of both curves (Fp2 coordinates; base-field multiplies out of line).
#define MSM_U_OUTLINE_PRODUCTS 1   // base-field products out of line (field_u.hpp)
#include "bucket_kernels.hpp"
namespace msm {
using F2a_ = Fp2<Bls12381Fp>;
using F2b_ = Fp2<Bn254Fp>;
MSM_INSTANTIATE_BUCKET_KERNELS(F2a_)
MSM_INSTANTIATE_BUCKET_KERNELS(F2b_)
}  // namespace msm
