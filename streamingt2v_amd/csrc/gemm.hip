// bf16 MFMA GEMM family for gfx950:  C[M,N] = epilogue( A_view[M,K] . W[N,K]^T ), fp32 accumulate.
//
// One kernel template serves every dense contraction of the StreamingSVD UNet / ControlNet / VAE:
//   * nn.Linear (attention.py, video_attention.py, conditioning.py of the reference),
//   * 3x3 convolutions as implicit GEMM on channels-last activations (K = 9*Cin, zero padding, stride 1|2,
//     optional nearest-2x upsample folded into the source addressing),
//   * the (3,1,1) temporal convolutions as a 3-tap implicit GEMM over the frame axis (no transposes).
//
// Design (CDNA4):
//   - v_mfma_f32_32x32x16_bf16; a wave owns a (FM*32) x (FN*32) output tile; the accumulator is kept
//     TRANSPOSED (D[i=n][j=m] = sum_k W[n][k] X[m][k]) so that a lane holds 4 consecutive output channels of
//     one token row -> 8-byte bf16 stores / 16-byte fp32 stores along the contiguous channel axis.
//   - A and W tiles are staged HBM -> LDS with global_load_lds (16 B / lane, no VGPR round trip).  The LDS image
//     is lane-linear, so the bank-conflict swizzle is applied on the per-lane SOURCE address and again on the
//     ds_read_b128 side (same involution): physical 16-B slot = logical k-slot ^ ((row >> 1) & 7).
//   - double-buffered LDS, one barrier per K tile (stage k+1 is issued before the MFMAs of tile k).
//   - XCD-aware tile order: each XCD walks a contiguous range of tiles, N fastest, so an A row-panel is
//     fetched from HBM once and the (small) weight matrix stays in L2 / Infinity Cache.
#include "svd_common.h"

namespace {

template <int BM_, int BN_, int WM_, int WN_, int BK_, bool GLDS_, bool TRANS_>
struct GemmCfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;
    static constexpr bool GLDS = GLDS_, TRANS = TRANS_;
    static constexpr int NT = WM * WN * 64;
    static constexpr int FM = BM / WM / 32, FN = BN / WN / 32;
    static constexpr int ROWB = BK * 2;       // bytes per LDS row
    static constexpr int SLOTS = BK / 8;      // 16-byte slots per row
    static constexpr int SW_SHIFT = (BK == 64) ? 1 : 2;
    static constexpr int RPP = NT / SLOTS;    // tile rows staged per pass
    static constexpr int PA = BM / RPP, PB = BN / RPP;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of rows-per-pass");
    static_assert(RPP % 16 == 0, "swizzle assumes pass stride multiple of 16 rows");
};

template <class CFG, int AMODE>
__global__ __launch_bounds__(CFG::NT) void gemm_kernel(const svd_gemm_args p) {
    constexpr int BM = CFG::BM, BN = CFG::BN, WM = CFG::WM, WN = CFG::WN, BK = CFG::BK;
    constexpr int NT = CFG::NT, FM = CFG::FM, FN = CFG::FN, ROWB = CFG::ROWB, SLOTS = CFG::SLOTS;
    constexpr int RPP = CFG::RPP, PA = CFG::PA, PB = CFG::PB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_base = lds_addr_of(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // ---- tile assignment (XCD-aware, bijective) ----
    const int tilesN = (p.N + BN - 1) / BN;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = wgid / tilesN, tile_n = wgid - tile_m * tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging coordinates ----
    const int srow = tid / SLOTS;                 // row within a pass
    const int ps = tid % SLOTS;                   // physical 16-B slot
    const int ls = ps ^ ((srow >> CFG::SW_SHIFT) & (SLOTS - 1));  // logical k-slot this lane fetches
    const int kofs = ls * 8;

    const svd_bf16* __restrict__ Ap = p.A + kofs;
    const svd_bf16* __restrict__ Zp = p.zeros + kofs;
    const uint32_t lda = (uint32_t)p.lda;

    // per staged row: element offset of the source row (32-bit: every activation tensor is < 2^32 elements)
    uint32_t a_off[PA];
    int a_y[PA], a_x[PA];   // conv: yo*stride-1, xo*stride-1 ; temporal: a_y = frame index t
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int m = m0 + j * RPP + srow;
        if (m > p.M - 1) m = p.M - 1;
        if constexpr (AMODE == SVD_A_PLAIN) {
            a_off[j] = (uint32_t)m * lda; a_y[j] = 0; a_x[j] = 0;
        } else if constexpr (AMODE == SVD_A_CONV3X3) {
            const int hw = p.hout * p.wout;
            const int f = m / hw, rem = m - f * hw;
            const int yo = rem / p.wout, xo = rem - yo * p.wout;
            a_off[j] = (uint32_t)(f * p.hin * p.win);   // first source pixel of the frame
            a_y[j] = yo * p.stride - 1; a_x[j] = xo * p.stride - 1;
        } else {
            const int fr = m / p.rows_per_frame;
            a_off[j] = (uint32_t)m * lda; a_y[j] = fr % p.t_frames; a_x[j] = 0;
        }
    }
    const svd_bf16* w_row[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        int n = n0 + j * RPP + srow;
        if (n > p.N - 1) n = p.N - 1;
        w_row[j] = p.W + (int64_t)n * p.ldw + kofs;
    }
    const int hlim = p.hin << p.ups, wlim = p.win << p.ups;
    const int ups = p.ups, win = p.win, t_frames = p.t_frames;
    const uint32_t frame_step = (uint32_t)p.rows_per_frame * lda;

    f32x16_t acc[FN][FM];
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    uint4 ra[CFG::GLDS ? 1 : PA], rb[CFG::GLDS ? 1 : PB];

    // running K position: k0 = tap * cin + c0 (tap / c0 only meaningful for the implicit-GEMM views)
    int k_next = 0, tap_next = 0, c_next = 0;

    auto a_src = [&](int j, int k0, int tap, int c0) -> const svd_bf16* {
        if constexpr (AMODE == SVD_A_PLAIN) {
            return Ap + ((size_t)a_off[j] + (uint32_t)k0);
        } else if constexpr (AMODE == SVD_A_CONV3X3) {
            const int dy = (tap * 11) >> 5;            // tap / 3 for tap in 0..8
            const int dx = tap - dy * 3;
            const int yu = a_y[j] + dy, xu = a_x[j] + dx;
            const bool ok = ((unsigned)yu < (unsigned)hlim) && ((unsigned)xu < (unsigned)wlim);
            const uint32_t pix = a_off[j] + (uint32_t)((yu >> ups) * win + (xu >> ups));
            return ok ? Ap + ((size_t)(pix * lda) + (uint32_t)c0) : Zp;
        } else {
            const int tt = a_y[j] + tap - 1;
            const bool ok = (unsigned)tt < (unsigned)t_frames;
            const uint32_t off = a_off[j] + (uint32_t)(tap - 1) * frame_step;   // wraps correctly mod 2^32
            return ok ? Ap + ((size_t)off + (uint32_t)c0) : Zp;
        }
    };

    auto issue = [&](int buf) {
        const int k0 = k_next, tap = tap_next, c0 = c_next;
        k_next += BK;
        if constexpr (AMODE != SVD_A_PLAIN) {
            c_next += BK;
            if (c_next >= p.cin) { c_next = 0; ++tap_next; }
        }
        if constexpr (CFG::GLDS) {
            const uint32_t dA = __builtin_amdgcn_readfirstlane(smem_base + buf * CFG::STAGE_BYTES + wave * 1024);
            const uint32_t dB = dA + CFG::A_BYTES;
#pragma unroll
            for (int j = 0; j < PA; ++j) glds16_asm(a_src(j, k0, tap, c0), dA + j * (NT * 16));
#pragma unroll
            for (int j = 0; j < PB; ++j) glds16_asm(w_row[j] + k0, dB + j * (NT * 16));
        } else {
#pragma unroll
            for (int j = 0; j < PA; ++j) ra[j] = *(const uint4*)a_src(j, k0, tap, c0);
#pragma unroll
            for (int j = 0; j < PB; ++j) rb[j] = *(const uint4*)(w_row[j] + k0);
        }
    };
    auto commit = [&](int buf) {   // register-staged path only
        if constexpr (!CFG::GLDS) {
            char* sA = smem + buf * CFG::STAGE_BYTES;
            char* sB = sA + CFG::A_BYTES;
#pragma unroll
            for (int j = 0; j < PA; ++j) *(uint4*)(sA + j * (NT * 16) + tid * 16) = ra[j];
#pragma unroll
            for (int j = 0; j < PB; ++j) *(uint4*)(sB + j * (NT * 16) + tid * 16) = rb[j];
        }
    };

    const int arow0 = wm * (BM / WM) + l31;   // + fm*32
    const int brow0 = wn * (BN / WN) + l31;   // + fn*32
    auto compute = [&](int buf) {
        const char* sA = smem + buf * CFG::STAGE_BYTES;
        const char* sB = sA + CFG::A_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int lslot = 2 * ks + hi;
            uint4 wf[FN], xf[FM];
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                const int row = brow0 + fn * 32;
                wf[fn] = *(const uint4*)(sB + row * ROWB + ((lslot ^ ((row >> CFG::SW_SHIFT) & (SLOTS - 1))) << 4));
            }
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int row = arow0 + fm * 32;
                xf[fm] = *(const uint4*)(sA + row * ROWB + ((lslot ^ ((row >> CFG::SW_SHIFT) & (SLOTS - 1))) << 4));
            }
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    if constexpr (CFG::TRANS)
                        acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, xf[fm]), __builtin_bit_cast(bf16x8_t, wf[fn]), acc[fn][fm], 0, 0, 0);
                    else
                        acc[fn][fm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8_t, wf[fn]), __builtin_bit_cast(bf16x8_t, xf[fm]), acc[fn][fm], 0, 0, 0);
                }
        }
    };

    // ---- main loop ----
    const int nk = p.K / BK;
    issue(0);
    commit(0);
    if constexpr (CFG::GLDS) svd_wait_dma();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1);
        compute(cur);
        if (kt + 1 < nk) commit(cur ^ 1);
        if constexpr (CFG::GLDS) svd_wait_dma();
        __syncthreads();
    }

    // ---- epilogue ----
    if constexpr (CFG::TRANS) {
        // acc[fn][fm] : D[i = m][j = n] ; lane -> n, regs -> 4 consecutive token rows
        svd_bf16* Cb = (svd_bf16*)p.C;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int n = n0 + wn * (BN / WN) + fn * 32 + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int fm = 0; fm < FM; ++fm)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + wm * (BM / WM) + fm * 32 + 8 * g + 4 * hi;
                    if (m >= p.M) continue;
                    const int f = m / p.tok_per_frame, tok = m - f * p.tok_per_frame;
                    uint2 o;
                    o.x = pack_bf16x2(acc[fn][fm][4 * g + 0] + bv, acc[fn][fm][4 * g + 1] + bv);
                    o.y = pack_bf16x2(acc[fn][fm][4 * g + 2] + bv, acc[fn][fm][4 * g + 3] + bv);
                    *(uint2*)(Cb + ((int64_t)f * p.N + n) * p.tokens_ld + tok) = o;
                }
        }
    } else {
        const bool geglu = (p.epi_flags & SVD_EPI_GEGLU) != 0;
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = m0 + wm * (BM / WM) + fm * 32 + l31;
            if (m >= p.M) continue;
            const float* rv = p.rowvec ? p.rowvec + (int64_t)(m / p.rows_per_vec) * p.rowvec_ld : nullptr;
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) {
                if (geglu && (fn & 1)) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nloc = wn * (BN / WN) + fn * 32 + 8 * g + 4 * hi;
                    const int nb = n0 + nloc;
                    if (nb >= p.N) continue;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[fn][fm][4 * g + i];
                    if (p.bias) {
                        const float4 b4 = *(const float4*)(p.bias + nb);
                        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                    }
                    int ncol = nb;
                    if (geglu) {
                        if constexpr (FN >= 2) {
                            float gt[4];
                            const int fg = (fn + 1 < FN) ? fn + 1 : fn;
#pragma unroll
                            for (int i = 0; i < 4; ++i) gt[i] = acc[fg][fm][4 * g + i];
                            if (p.bias) {
                                const float4 b4 = *(const float4*)(p.bias + nb + 32);
                                gt[0] += b4.x; gt[1] += b4.y; gt[2] += b4.z; gt[3] += b4.w;
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] *= gelu_erf_f(gt[i]);
                        }
                        ncol = ((n0 + wn * (BN / WN) + fn * 32) >> 1) + 8 * g + 4 * hi;
                    }
                    if (rv) {
                        const float4 r4 = *(const float4*)(rv + ncol);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                    if (p.R) {
                        const uint2 r2 = *(const uint2*)(p.R + (int64_t)m * p.ldr + ncol);
                        v[0] += bf16lo_to_f32(r2.x); v[1] += bf16hi_to_f32(r2.x);
                        v[2] += bf16lo_to_f32(r2.y); v[3] += bf16hi_to_f32(r2.y);
                    }
                    if (p.S) {
                        const uint2 s2 = *(const uint2*)(p.S + (int64_t)m * p.lds + ncol);
                        const float a = p.alpha, b = 1.f - p.alpha;
                        v[0] = a * bf16lo_to_f32(s2.x) + b * v[0]; v[1] = a * bf16hi_to_f32(s2.x) + b * v[1];
                        v[2] = a * bf16lo_to_f32(s2.y) + b * v[2]; v[3] = a * bf16hi_to_f32(s2.y) + b * v[3];
                    }
                    if (p.epi_flags & SVD_EPI_SILU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = silu_f(v[i]);
                    }
                    if (p.out_mode == SVD_OUT_F32) {
                        *(float4*)((float*)p.C + (int64_t)m * p.ldc + ncol) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 o;
                        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                        *(uint2*)((svd_bf16*)p.C + (int64_t)m * p.ldc + ncol) = o;
                    }
                }
            }
        }
    }
}

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

template <class CFG, int AMODE>
int launch_mode(const svd_gemm_args& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<CFG, AMODE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS_BYTES);
        if (e != hipSuccess) { svd_set_error_("gemm: hipFuncSetAttribute", e); return SVD_ELAUNCH; }
        attr_set = true;
    }
    const int tilesM = (a.M + CFG::BM - 1) / CFG::BM, tilesN = (a.N + CFG::BN - 1) / CFG::BN;
    hipLaunchKernelGGL((gemm_kernel<CFG, AMODE>), dim3(tilesM * tilesN), dim3(CFG::NT), CFG::LDS_BYTES, s, a);
    SVD_CHECK_LAUNCH("gemm launch");
    return SVD_OK;
}
template <class CFG>
int launch(const svd_gemm_args& a, hipStream_t s) {
    if (a.a_mode == SVD_A_PLAIN) return launch_mode<CFG, SVD_A_PLAIN>(a, s);
    if constexpr (!CFG::TRANS) {
        if (a.a_mode == SVD_A_CONV3X3) return launch_mode<CFG, SVD_A_CONV3X3>(a, s);
        if (a.a_mode == SVD_A_TEMPORAL3) return launch_mode<CFG, SVD_A_TEMPORAL3>(a, s);
    }
    return SVD_EINVAL;
}

template <class CFG>
void info(int* bm, int* bn, int* thr, int* lds) {
    if (bm) *bm = CFG::BM; if (bn) *bn = CFG::BN; if (thr) *thr = CFG::NT; if (lds) *lds = CFG::LDS_BYTES;
}

// can config `cfg` run these arguments?  (tile-shape independent checks live in svd_gemm)
template <class CFG>
bool cfg_ok(const svd_gemm_args& a) {
    const int kq = (a.a_mode == SVD_A_PLAIN) ? a.K : a.cin;
    if (kq % CFG::BK != 0 || a.K % CFG::BK != 0) return false;
    if (CFG::TRANS != (a.out_mode == SVD_OUT_BF16_T)) return false;
    if ((a.epi_flags & SVD_EPI_GEGLU) && (CFG::FN % 2 != 0)) return false;   // value|gate frag pairs per wave
    return true;
}

int pick_cfg(const svd_gemm_args& a) {
    const int kq = (a.a_mode == SVD_A_PLAIN) ? a.K : a.cin;
    if (a.out_mode == SVD_OUT_BF16_T) return 7;
    if (kq % 64 != 0 || a.K % 64 != 0) return 5;
    if (a.epi_flags & SVD_EPI_GEGLU) return (a.M >= 65536) ? 2 : 1;
    if (a.N % 320 == 0 && a.N <= 960) return 4;
    if (a.N <= 64) return 3;
    if (a.M >= 65536 && a.N % 128 == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int svd_gemm_num_configs(void) { return kNumCfg; }
extern "C" int svd_gemm_pick_config(const svd_gemm_args* args) { return args ? pick_cfg(*args) : SVD_EINVAL; }

extern "C" int svd_gemm_config_info(int cfg, int* bm, int* bn, int* threads, int* lds_bytes) {
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr) case id: info<Cfg##id>(bm, bn, threads, lds_bytes); return SVD_OK;
        SVD_GEMM_CONFIGS(X)
#undef X
    }
    return SVD_EINVAL;
}

extern "C" int svd_gemm_config_valid(const svd_gemm_args* args, int cfg) {
    if (!args) return SVD_EINVAL;
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr) case id: return cfg_ok<Cfg##id>(*args) ? 1 : 0;
        SVD_GEMM_CONFIGS(X)
#undef X
    }
    return SVD_EINVAL;
}

extern "C" int svd_gemm(const svd_gemm_args* args, svd_stream_t stream) {
    if (!args) return SVD_EINVAL;
    const svd_gemm_args& a = *args;
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return SVD_EINVAL;
    if (a.N % 4 != 0 && a.out_mode != SVD_OUT_BF16_T) return SVD_EINVAL;
    if (a.lda % 8 != 0 || a.ldw % 8 != 0) return SVD_EINVAL;             // 16-byte staged loads
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return SVD_EINVAL;
    if (a.a_mode != SVD_A_PLAIN) {
        if (!a.zeros || a.cin <= 0 || a.cin % 32 != 0) return SVD_EINVAL;
        if (a.a_mode == SVD_A_CONV3X3) {
            if (a.K != 9 * a.cin || a.hin <= 0 || a.win <= 0 || a.hout <= 0 || a.wout <= 0) return SVD_EINVAL;
            if ((a.stride != 1 && a.stride != 2) || (a.ups != 0 && a.ups != 1)) return SVD_EINVAL;
            if (a.M % (a.hout * a.wout) != 0) return SVD_EINVAL;
        } else if (a.a_mode == SVD_A_TEMPORAL3) {
            if (a.K != 3 * a.cin || a.t_frames <= 0 || a.rows_per_frame <= 0) return SVD_EINVAL;
            if (a.M % (a.t_frames * a.rows_per_frame) != 0) return SVD_EINVAL;
        } else return SVD_EINVAL;
    }
    if (a.K % 32 != 0) return SVD_EINVAL;
    if (a.rowvec && a.rows_per_vec <= 0) return SVD_EINVAL;
    if (a.out_mode == SVD_OUT_BF16_T) {
        if (a.tok_per_frame <= 0 || a.tok_per_frame % 4 != 0 || a.M % a.tok_per_frame != 0 || a.tokens_ld % 4 != 0)
            return SVD_EINVAL;
        if (a.R || a.S || a.rowvec || a.epi_flags) return SVD_EINVAL;
    }
    if (a.epi_flags & SVD_EPI_GEGLU) {
        if (a.N % 64 != 0 || a.out_mode != SVD_OUT_BF16) return SVD_EINVAL;
    }
    const int cfg = a.tile_cfg > 0 ? a.tile_cfg : pick_cfg(a);
    if (svd_gemm_config_valid(args, cfg) != 1) return SVD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr) case id: return launch<Cfg##id>(a, s);
        SVD_GEMM_CONFIGS(X)
#undef X
    }
    return SVD_EINVAL;
}
