// The MFMA GEMM family with IEEE-half operands (dtype 2): hugs_gemm.hip compiled a second time, see the note at its top.
// Provides hugs_gemm_nt_impl_f16 / hugs_gemm_tn_impl_f16, which the C entry points in hugs_gemm.hip dispatch to.
#define HUGS_GEMM_F16 1
#include "hugs_gemm.hip"
