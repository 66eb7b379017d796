#define SVD_GEMM_ELEM ElemBF16
#define SVD_GEMM_LAUNCH_NAME svd_gemm_launch_bf16
#include "gemm_impl.inc"
