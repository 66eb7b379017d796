// part 0 of the GEMM tile table (gemm_cfg.h), f16 elements
#define SVD_GEMM_ELEM ElemF16
#define SVD_GEMM_LAUNCH_NAME svd_gemm_launch_f16_p0
#define SVD_GEMM_CONFIGS_TU(X) SVD_GEMM_CONFIGS_P0(X)
#include "gemm_impl.inc"
