// Tile configurations of the GEMM family (shared by the per-element-type translation units and the dispatcher).
#pragma once
#include "svd_common.h"

namespace svd_gemm_detail {


template <int BM_, int BN_, int WM_, int WN_, int BK_, bool GLDS_, bool TRANS_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;
    static constexpr bool GLDS = GLDS_, TRANS = TRANS_;
    static constexpr int NT = WM * WN * 64;
    // register budget: 4-wave workgroups with <= 64 accumulator registers per lane must fit twice per SIMD (2 WG / CU)
    static constexpr int MIN_WAVES_PER_SIMD = (WM * WN == 4 && (BM / WM / 32) * (BN / WN / 32) * 16 <= 64) ? 2 : 1;
    static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    static constexpr int ROWB = BK * 2;       // bytes per LDS row
    static constexpr int SLOTS = BK / 8;      // 16-byte slots per row
    static constexpr int SW_SHIFT = (BK == 64) ? 1 : 2;
    static constexpr int RPP = NT / SLOTS;    // tile rows staged per pass
    static constexpr int PA = BM / RPP, PB = BN / RPP;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    // row-contiguous epilogue: per-wave fp32 staging slice of one 32x32 fragment (row stride 36 floats)
    static constexpr int EPI_SROW = 32 + 4;
    static constexpr int EPI_WAVE_BYTES = 32 * EPI_SROW * 4;
    static constexpr bool EPI_DEDICATED = (WM * WN * EPI_WAVE_BYTES > STAGE_BYTES);   // else: reuse the consumed stage buffer
    static constexpr int LAUNCH_LDS = LDS_BYTES + (EPI_DEDICATED ? WM * WN * EPI_WAVE_BYTES : 0);
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of rows-per-pass");
    static_assert(RPP % 16 == 0, "swizzle assumes pass stride multiple of 16 rows");
};

// ---- configuration table ---------------------------------------------------------------------------------
//        id  BM   BN   WM WN BK  GLDS   TRANS
#define SVD_GEMM_CONFIGS(X)                                                                           \
    X(1, 128, 128, 2, 2, 64, true, false)  /* default: 4 waves, 64x64 per wave, 2 WG/CU            */ \
    X(2, 256, 128, 4, 2, 64, true, false)  /* 8 waves, 64x64 per wave                              */ \
    X(3, 128, 64, 2, 2, 64, true, false)   /* narrow N                                             */ \
    X(4, 128, 320, 2, 2, 64, true, false)  /* N = 320 / 960 exactly, 64x160 per wave               */ \
    X(5, 128, 128, 2, 2, 32, true, false)  /* K (or cin) multiple of 32 only                       */ \
    X(6, 128, 128, 2, 2, 64, false, false) /* register-staged fallback of 1                        */ \
    X(7, 128, 128, 2, 2, 64, true, true)   /* transposed output (V^T for attention)                */ \
    X(8, 256, 256, 4, 2, 64, true, false)  /* 8 waves, 64x128 per wave                             */ \
    X(9, 128, 256, 2, 2, 64, true, false)  /* 4 waves, 64x128 per wave                             */ \
    X(10, 256, 128, 2, 2, 64, true, false) /* 4 waves, 128x64 per wave                             */ \
    X(11, 256, 256, 2, 2, 64, true, false) /* 4 waves, 128x128 per wave (1 wave / SIMD)            */ \
    X(12, 64, 128, 2, 2, 64, true, false)  /* small M                                              */ \
    X(13, 128, 160, 4, 1, 64, true, false) /* N = 320 as 2 tiles, 32x160 per wave                  */ \
    X(14, 256, 160, 4, 1, 64, true, false) /* N = 320 as 2 tiles, 64x160 per wave                  */ \
    X(15, 256, 64, 4, 1, 64, true, false)  /* N = 320 as 5 tiles, 64x64 per wave                   */ \
    X(16, 128, 192, 2, 2, 64, true, false) /* N = 960 / 1920 / 3840 exactly, 64x96 per wave        */
constexpr int kNumCfg = 16;

#define X(id, bm, bn, wm, wn, bk, glds, tr) using Cfg##id = GemmCfg<bm, bn, wm, wn, bk, glds, tr>;
SVD_GEMM_CONFIGS(X)
#undef X


}  // namespace svd_gemm_detail

// one per element type, defined in gemm_bf16.hip / gemm_f16.hip
int svd_gemm_launch_bf16(const svd_gemm_args& a, int cfg, hipStream_t s);
int svd_gemm_launch_f16(const svd_gemm_args& a, int cfg, hipStream_t s);
