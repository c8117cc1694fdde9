// The same family with THREE bf16 planes per operand (bf16x6: the VAE family's fp32-grade math mode, round 6).  Interface: uad_d16s_launch_v3.
#define UAD_D16S_NPL 3
#include "uad_gemm_d16s_body.inc"
