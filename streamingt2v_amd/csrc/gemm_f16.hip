#define SVD_GEMM_ELEM ElemF16
#define SVD_GEMM_LAUNCH_NAME svd_gemm_launch_f16
#include "gemm_impl.inc"
