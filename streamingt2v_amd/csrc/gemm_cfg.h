// Tile configurations of the GEMM family (shared by the per-element-type translation units and the dispatcher).
#pragma once
#include "svd_common.h"

#ifndef SVD_GEMM_BK32_WAVES
#define SVD_GEMM_BK32_WAVES 2   /* waves per SIMD that the 4-wave, 64-accumulator, BK 32 configurations are compiled for */
#endif

namespace svd_gemm_detail {


// internal A-operand view: 3x3 convolution whose source is read through a folded nearest-2x upsample (svd_gemm_args.ups = 1)
#define SVD_A_CONV3X3_UPS 3

#ifndef SVD_GEMM_FRAG_SETS_MAX
#define SVD_GEMM_FRAG_SETS_MAX 2      /* 1: A/B switch, single fragment set, scheduler's own order */
#endif
template <int BM_, int BN_, int WM_, int WN_, int BK_, bool GLDS_, bool TRANS_, int NS_ = 2, bool DELAY_ = false, bool PP_ = false>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;
    static constexpr bool GLDS = GLDS_, TRANS = TRANS_;
    static constexpr int NT = WM * WN * 64;        // threads of one wave GROUP (= the workgroup unless ping-pong)
    static constexpr bool PINGPONG = PP_ && !TRANS_;
    static constexpr int GROUPS = PINGPONG ? 2 : 1;  // ping-pong: two independent wave groups per workgroup, half an iteration apart
    static constexpr int THREADS = NT * GROUPS;
    // register budget: 4-wave workgroups with <= 64 accumulator registers per lane must fit twice per SIMD (2 WG / CU)
    static constexpr int MIN_WAVES_PER_SIMD = (!PP_ && WM * WN == 4 && ((BM / WM / 32) * (BN / WN / 32) * 16 <= 64 || BK_ == 32)) ? ((BK_ == 32 && (BM / WM / 32) * (BN / WN / 32) * 16 <= 64) ? SVD_GEMM_BK32_WAVES : 2) : 1;   // BK 32 + 128 accumulators: 2 x 58 KB LDS, 256 registers
    static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    // tiles that also carry the folded-upsample convolution kernel (the Upsample layers of the UNet and of the VAE decoder): the heuristic's
    // picks and the shapes the tuner has chosen for those layers -- not all 22, to bound build time
    static constexpr bool UPS_KERNEL = !TRANS_ && !PP_ && ((BM_ == 128 && (BN_ == 128 || BN_ == 64 || BN_ == 320) && WM_ == 2 && NS_ == 2) || (BM_ == 256 && (BN_ == 128 || BN_ == 256 || BN_ == 320) && WM_ * WN_ == 8 && BK_ == 64));
    // tiles that also carry the fp32-residual-stream kernel (gemm_impl.inc ST = true): the heuristic's picks and the tiles the tuner chooses for
    // the stream's producers -- not all 23, to bound build time.  ids 1 2 3 4 5 8 12 14 18 20 21 22
    static constexpr bool STREAM_KERNEL = !TRANS_ && !PP_ &&
        ((BM_ == 128 && BN_ == 128 && WM_ == 2 && NS_ == 2 && GLDS_) || (BM_ == 256 && BN_ == 128 && WM_ == 4 && BK_ == 64) ||
         (BM_ == 128 && BN_ == 64) || (BM_ == 128 && BN_ == 320) || (BM_ == 256 && BN_ == 256 && BK_ == 64 && WM_ * WN_ == 8) ||
         (BM_ == 64 && BN_ == 128) || (BM_ == 256 && BN_ == 160) || (BM_ == 256 && BN_ == 320));
    // fragment register sets of the k-step software pipeline (gemm_impl.inc compute()): two when accumulators + 2 x fragments leave
    // ~70 registers for addresses / epilogue state inside the wave's share of the 512-entry register file.
    static constexpr int WAVES_PER_SIMD = (THREADS / 256 > MIN_WAVES_PER_SIMD) ? THREADS / 256 : MIN_WAVES_PER_SIMD;
    static constexpr int REG_BUDGET = 512 / WAVES_PER_SIMD;
    static constexpr int FRAG_SETS = (SVD_GEMM_FRAG_SETS_MAX >= 2 && FM * FN * 16 + 2 * (FM + FN) * 4 + 70 <= REG_BUDGET) ? 2 : 1;
    static constexpr int ROWB = BK * 2;       // bytes per LDS row
    static constexpr int SLOTS = BK / 8;      // 16-byte slots per row
    static constexpr int SW_SHIFT = (BK == 64) ? 1 : 2;
    static constexpr int RPP = NT / SLOTS;    // tile rows staged per pass
    static constexpr int PA = BM / RPP, PB = BN / RPP;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int NSTAGE = NS_;        // LDS ring depth: the load stream runs NSTAGE-1 K tiles ahead of the MFMAs
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
    // row-contiguous epilogue: per-wave fp32 staging slice of one 32x32 fragment (row stride 36 floats)
    static constexpr int EPI_SROW = 32 + 4;
    static constexpr int EPI_WAVE_BYTES = 32 * EPI_SROW * 4;
    // delayed epilogue: the finished tile's passes run inside the next tile's K loop (needs a 2nd accumulator set: <= 64 regs)
    static constexpr bool DELAYED_EPI = DELAY_ && !TRANS_ && ((BM / WM / 32) * (BN / WN / 32) * 16 <= 64);
    static constexpr bool EPI_DEDICATED = DELAYED_EPI || (WM * WN * EPI_WAVE_BYTES > STAGE_BYTES);   // else: reuse the consumed stage buffer
    // aux slots: bias slice + per-frame vectors of up to AUX_NRV frames, DMA'd at tile setup (2 slots: current / prefetched tile)
    static constexpr int AUX_NRV = (2 * (BM_ + BN_) * BK_ * 2 >= 144 * 1024) ? 2 : 4;   // fewer per-frame vector rows when the stage ring leaves < 16 KB
    static constexpr int AUX_INSTR = (BN + 255) / 256;
    static constexpr int AUX_SLOT_BYTES = (1 + AUX_NRV) * AUX_INSTR * 1024;
    static constexpr int AUX_OFF = LDS_BYTES + (EPI_DEDICATED ? WM * WN * EPI_WAVE_BYTES : 0);
    static constexpr int AUX_SLOTS = NSTAGE + (DELAYED_EPI ? 1 : 0);   // one per tile the load stream can be ahead (+1: the delayed tile)
    static constexpr int LAUNCH_LDS = AUX_OFF + (TRANS_ ? 0 : AUX_SLOTS * AUX_SLOT_BYTES);
    static constexpr int TOTAL_LDS = GROUPS * LAUNCH_LDS;            // LAUNCH_LDS: bytes of ONE group
    static_assert(TOTAL_LDS <= 160 * 1024, "tile configuration exceeds the 160 KiB LDS of a CU");
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of rows-per-pass");
    static_assert(RPP % 16 == 0, "swizzle assumes pass stride multiple of 16 rows");
};

// ---- configuration table ---------------------------------------------------------------------------------
//        id  BM   BN   WM WN BK  GLDS   TRANS  NSTAGE
// The table is split into four parts: every part is instantiated in its own pair of translation units (gemm_{bf16,f16}_p0..3.hip), so that the
// 23 tiles x 4 views x 2 element types compile on 8 cores in parallel (one pair of units took 12 minutes).
#ifndef SVD_GEMM_CONFIGS   /* a probe build may predefine a reduced table: it goes to part 0 */
#define SVD_GEMM_CONFIGS_P0(X)                                                            \
    X(1, 128, 128, 2, 2, 64, true, false, 2)       /* default: 4 waves, 64x64 per wave, 2 WG/CU            */ \
    X(5, 128, 128, 2, 2, 32, true, false, 2)       /* K (or cin) multiple of 32 only                       */ \
    X(9, 128, 256, 2, 2, 64, true, false, 2)       /* 4 waves, 64x128 per wave                             */ \
    X(13, 128, 160, 4, 1, 64, true, false, 2)      /* N = 320 as 2 tiles, 32x160 per wave                  */ \
    X(17, 256, 256, 4, 2, 32, true, false, 3)      /* 3-stage ring, BK 32: loads 2 K tiles ahead (133 KB)          */ \
    X(21, 256, 320, 4, 2, 64, true, false, 2)      /* N = 320 k exactly (320 / 640 / 960 / 1280 ...), 8 waves, 64x160 per wave (157 KB)     */
#define SVD_GEMM_CONFIGS_P1(X)                                                            \
    X(2, 256, 128, 4, 2, 64, true, false, 2)       /* 8 waves, 64x64 per wave                              */ \
    X(6, 128, 128, 2, 2, 64, false, false, 2)      /* register-staged fallback of 1                        */ \
    X(10, 256, 128, 2, 2, 64, true, false, 2)      /* 4 waves, 128x64 per wave                             */ \
    X(14, 256, 160, 4, 1, 64, true, false, 2)      /* N = 320 as 2 tiles, 64x160 per wave                  */ \
    X(18, 256, 128, 4, 2, 64, true, false, 3)      /* 3-stage ring (144 KB)                                        */ \
    X(22, 128, 320, 4, 2, 64, true, false, 2)      /* same, 32x160 per wave                                                                 */
#define SVD_GEMM_CONFIGS_P2(X)                                                            \
    X(3, 128, 64, 2, 2, 64, true, false, 2)        /* narrow N                                             */ \
    X(7, 128, 128, 2, 2, 64, true, true, 2)        /* transposed output (V^T for attention)                */ \
    X(11, 256, 256, 2, 2, 64, true, false, 2)      /* 4 waves, 128x128 per wave (1 wave / SIMD)            */ \
    X(15, 256, 64, 4, 1, 64, true, false, 2)       /* N = 320 as 5 tiles, 64x64 per wave                   */ \
    X(19, 256, 128, 4, 2, 32, true, false, 4)      /* 4-stage ring, BK 32: requests 3 K tiles ahead (133 KB)                                  */ \
    X(23, 128, 256, 2, 2, 32, true, false, 2)      /* 4 waves, 64x128 per wave, BK 32: 58 KB -> TWO workgroups per CU (one's epilogue under the other's MFMAs) */
#define SVD_GEMM_CONFIGS_P3(X)                                                            \
    X(4, 128, 320, 2, 2, 64, true, false, 2)       /* N = 320 / 960 exactly, 64x160 per wave               */ \
    X(8, 256, 256, 4, 2, 64, true, false, 2)       /* 8 waves, 64x128 per wave                             */ \
    X(12, 64, 128, 2, 2, 64, true, false, 2)       /* small M                                              */ \
    X(16, 128, 192, 2, 2, 64, true, false, 2)      /* N = 960 / 1920 / 3840 exactly, 64x96 per wave        */ \
    X(20, 256, 256, 2, 4, 64, true, false, 2)      /* 8 with 128x64 wave tiles                                                              */
#define SVD_GEMM_CONFIGS(X) SVD_GEMM_CONFIGS_P0(X) SVD_GEMM_CONFIGS_P1(X) SVD_GEMM_CONFIGS_P2(X) SVD_GEMM_CONFIGS_P3(X)
#else
#define SVD_GEMM_CONFIGS_P0(X) SVD_GEMM_CONFIGS(X)
#define SVD_GEMM_CONFIGS_P1(X)
#define SVD_GEMM_CONFIGS_P2(X)
#define SVD_GEMM_CONFIGS_P3(X)
#endif
constexpr int kNumCfg = 23;

#define X(id, bm, bn, wm, wn, bk, glds, tr, ns) using Cfg##id = GemmCfg<bm, bn, wm, wn, bk, glds, tr, (ns) % 10, (((ns) / 10) & 1) != 0, (((ns) / 10) & 2) != 0>;   /* tens digit: 1 = delayed epilogue, 2 = ping-pong */
SVD_GEMM_CONFIGS(X)
#undef X


}  // namespace svd_gemm_detail

// one per element type, defined in gemm_bf16.hip / gemm_f16.hip
// m_base (round 6): first row of the launch's window -- rows [m_base, a.M) of the problem are computed, all addressing stays absolute (the tail of a launch whose
// last round would be nearly empty runs as a second launch with a smaller tile: svd_gemm_tail_launch_, gemm.hip)
#define SVD_GEMM_DECL_PART(P) int svd_gemm_launch_bf16_p##P(const svd_gemm_args& a, int cfg, hipStream_t s, int m_base); int svd_gemm_launch_f16_p##P(const svd_gemm_args& a, int cfg, hipStream_t s, int m_base);
SVD_GEMM_DECL_PART(0) SVD_GEMM_DECL_PART(1) SVD_GEMM_DECL_PART(2) SVD_GEMM_DECL_PART(3)
#undef SVD_GEMM_DECL_PART
// rows [m_base, a.M) with a tile smaller than `main_bm` rows (same bits: every tile configuration accumulates an output element in the same order); SVD_EINVAL if none fits
int svd_gemm_tail_launch_(const svd_gemm_args& a, int main_bm, int m_base, hipStream_t s);
