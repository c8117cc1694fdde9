// The lane = pixel D-kind k5 s2 kernels (uad_conv16s.inc), bf16x3 products, as a translation unit of their own (the family's fully unrolled instances used to
// double uad_gemm.hip's compile time).  Interface: uad_d16s_takes_v / uad_d16s_launch_v2 (declared in uad_gemm.hip).
#define UAD_D16S_NPL 2
#include "uad_gemm_d16s_body.inc"

bool uad_d16s_takes_v(const void* conv_gemm_args) { return conv5_d16s_takes(*static_cast<const ConvGemmArgs*>(conv_gemm_args)); }
