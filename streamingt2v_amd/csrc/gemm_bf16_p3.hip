// part 3 of the GEMM tile table (gemm_cfg.h), bf16 elements
#define SVD_GEMM_ELEM ElemBF16
#define SVD_GEMM_LAUNCH_NAME svd_gemm_launch_bf16_p3
#define SVD_GEMM_CONFIGS_TU(X) SVD_GEMM_CONFIGS_P3(X)
#include "gemm_impl.inc"
