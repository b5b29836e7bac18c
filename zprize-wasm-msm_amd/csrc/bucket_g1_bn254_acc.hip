// Bucket ACCUMULATION kernel for BN254 G1: everything inline, one dependent chain of multiply-adds per field product.
#define MSM_U_CHAINED_MADS 1
#include "bucket_kernels.hpp"
namespace msm {
using F_ = Fp<Bn254Fp, true>;
MSM_INSTANTIATE_ACCUMULATE(F_, 1)
MSM_INSTANTIATE_SLICES(F_, 1)
MSM_INSTANTIATE_ITEMS(F_, 1)
}  // namespace msm
