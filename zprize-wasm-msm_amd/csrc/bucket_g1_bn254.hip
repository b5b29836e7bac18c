// Bucket kernels for BN254 G1 (254-bit Fp, multiplies expanded inline).
#include "bucket_kernels.hpp"
namespace msm {
using F_ = Fp<Bn254Fp, true>;
MSM_INSTANTIATE_BUCKET_KERNELS(F_)
}  // namespace msm
