// Extended-precision "rim" of the denoiser (round 4).
//
// oracle/ablate_precision_sites.py (CPU emulation of the 16-bit execution, shipped architecture): 15 % of the squared deviation of
// StreamingWrapper.forward from the reference's fp32 path enters through the ControlNet's image-condition embedding (eight convolutions at
// up to 576x1024 pixels with 3..512 channels, controlnet.py:51-121 -- computed ONCE per chunk here), 7.5 % through the UNet's head
// (out.0 GroupNorm + SiLU + out.2 conv 320 -> 4, video_model.py:493-494,617), 3-4 % through the two stem convolutions.  These sites are a
// fraction of a percent of the job's FLOPs, so they run with ~22-bit operands instead of 11-bit ones:
//   * SPLIT-3 GEMM operands.  A = A_hi + A_lo and W = W_hi + W_lo in the 16-bit element type (x_hi = rn16(x), x_lo = rn16(x - x_hi)); the
//     ordinary MFMA GEMM / implicit-GEMM convolution then runs on A' = [A_hi | A_lo | A_hi], W' = [W_hi | W_hi | W_lo] (K tripled, per tap for
//     the convolution views) and accumulates A_hi W_hi + A_lo W_hi + A_hi W_lo in fp32: everything but the 2^-22 cross term.
//     svd_rows_split3 writes A' from fp32 rows (optionally through the per-pixel LayerNorm + SiLU of the embedding, in fp32);
//     svd_nchw_to_tokens_x3 writes it from the NCHW fp32 inputs of the stems.
//   * the head as ONE fp32 kernel (svd_head_gn_silu_conv3x3): GroupNorm apply + SiLU + the 3x3 convolution to <= 4 channels on the VALU --
//     with 4 output channels an MFMA tile is 94 % padding (the round-3 trace has this GEMM at 34 TFLOP/s, 1.28 ms per forward).
#include "svd_common.h"

namespace {

template <class E> __device__ __forceinline__ void split_hi_lo(const float (&v)[8], uint4& hi, uint4& lo) {
    float r[8];
    uint32_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = E::pack(v[2 * i], v[2 * i + 1]);
        r[2 * i] = v[2 * i] - E::lo(h[i]);
        r[2 * i + 1] = v[2 * i + 1] - E::hi(h[i]);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(E::pack(r[0], r[1]), E::pack(r[2], r[3]), E::pack(r[4], r[5]), E::pack(r[6], r[7]));
}

// NCHW fp32 (X0 * scale[frame] | X1 on the channel axis, zero-padded to cpad) -> rows [hi(cpad) | lo(cpad) | hi(cpad)]
template <class E>
__global__ void nchw_to_tokens_x3_kernel(const float* __restrict__ X0, int c0, const float* __restrict__ X1, int c1,
                                         const float* __restrict__ scale, svd_bf16* __restrict__ Y, int cpad, int frames, int pix) {
    const int64_t total = (int64_t)frames * pix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / pix), p = (int)(i - (int64_t)f * pix);
        const float sc = scale ? scale[f] : 1.f;
        svd_bf16* y = Y + i * 3 * cpad;
        for (int cb = 0; cb < cpad; cb += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = cb + k;
                float x = 0.f;
                if (c < c0) x = X0[((int64_t)f * c0 + c) * pix + p] * sc;
                else if (c < c0 + c1) x = X1[((int64_t)f * c1 + (c - c0)) * pix + p];
                v[k] = x;
            }
            uint4 hi, lo;
            split_hi_lo<E>(v, hi, lo);
            *(uint4*)(y + cb) = hi;
            *(uint4*)(y + cpad + cb) = lo;
            *(uint4*)(y + 2 * cpad + cb) = hi;
        }
    }
}

// fp32 rows [rows, C] -> split-3 rows [rows, 3 C]; optional LayerNorm over the C channels of a row (fp32, two passes over registers) and
// SiLU first.  LPR lanes per row, one octet per lane (C <= 8 * LPR): the embedding's rows have 32 .. 512 channels.
template <int LPR, class E>
__global__ __launch_bounds__(256) void rows_split3_kernel(const float* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y, int64_t ldy,
                                                          int64_t rows, int channels, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int do_ln, int do_silu) {
    constexpr int RW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int octets = channels >> 3;
    const float invc = 1.f / (float)channels;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const bool has = sub < octets;
    float gg[8], bb[8];
    if (do_ln && has) {
        const float4 g0 = *(const float4*)(gamma + sub * 8), g1 = *(const float4*)(gamma + sub * 8 + 4);
        const float4 b0 = *(const float4*)(beta + sub * 8), b1 = *(const float4*)(beta + sub * 8 + 4);
        gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    if (octets > LPR) {
        // wide rows without LayerNorm (the embedding MLPs' 768 / 1280-wide activations): no row statistics, so every octet is independent
        for (int64_t row0 = wave_id * RW; row0 < rows; row0 += nwaves * RW) {
            const int64_t row = row0 + rsel;
            if (row >= rows) continue;
            for (int o = sub; o < octets; o += LPR) {
                const float* xp = X + row * ldx + o * 8;
                const float4 a = *(const float4*)xp, b = *(const float4*)(xp + 4);
                float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                if (do_silu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = v[i] / (1.0f + __expf(-v[i]));
                }
                uint4 hi, lo;
                split_hi_lo<E>(v, hi, lo);
                svd_bf16* y = Y + row * ldy + o * 8;
                *(uint4*)y = hi;
                *(uint4*)(y + channels) = lo;
                *(uint4*)(y + 2 * channels) = hi;
            }
        }
        return;
    }
    for (int64_t row0 = wave_id * RW; row0 < rows; row0 += nwaves * RW) {
        const bool live = row0 + rsel < rows;
        const int64_t row = live ? row0 + rsel : rows - 1;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        if (has) {
            const float* xp = X + row * ldx + sub * 8;
            const float4 a = *(const float4*)xp, b = *(const float4*)(xp + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        }
        if (do_ln) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mean = s * invc;
            float sq = 0.f;
            if (has) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; sq += d * d; }
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = rsqrtf(sq * invc + eps);
            if (has) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean) * rstd * gg[i] + bb[i];
            }
        }
        if (do_silu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = v[i] / (1.0f + __expf(-v[i]));      // IEEE division: this path is about precision, not rate
        }
        if (has && live) {
            uint4 hi, lo;
            split_hi_lo<E>(v, hi, lo);
            svd_bf16* y = Y + row * ldy + sub * 8;
            *(uint4*)y = hi;
            *(uint4*)(y + channels) = lo;
            *(uint4*)(y + 2 * channels) = hi;
        }
    }
}

// Y = X + B with B fp32 (the image-condition embedding added to the ControlNet's stem output, controlnet.py:23-48): the sum is formed in fp32
// and rounded once to X's type (16-bit stream) or kept fp32 (fp32 residual stream).
template <class E, bool XF32>
__global__ void add_rows_bf32_kernel(const void* __restrict__ X, int64_t ldx, const float* __restrict__ B, int64_t ldb, void* __restrict__ Y,
                                     int64_t ldy, int64_t rows, int c) {
    const int oct = c >> 3;
    const int64_t total = rows * oct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / oct;
        const int o = (int)(i - r * oct);
        const float* bp = B + r * ldb + o * 8;
        const float4 b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 4);
        if constexpr (XF32) {
            const float* xp = (const float*)X + r * ldx + o * 8;
            const float4 a0 = *(const float4*)xp, a1 = *(const float4*)(xp + 4);
            float* yp = (float*)Y + r * ldy + o * 8;
            *(float4*)yp = make_float4(a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w);
            *(float4*)(yp + 4) = make_float4(a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w);
        } else {
            const uint4 a = *(const uint4*)((const svd_bf16*)X + r * ldx + o * 8);
            uint4 w;
            w.x = E::pack(E::lo(a.x) + b0.x, E::hi(a.x) + b0.y); w.y = E::pack(E::lo(a.y) + b0.z, E::hi(a.y) + b0.w);
            w.z = E::pack(E::lo(a.z) + b1.x, E::hi(a.z) + b1.y); w.w = E::pack(E::lo(a.w) + b1.z, E::hi(a.w) + b1.w);
            *(uint4*)((svd_bf16*)Y + r * ldy + o * 8) = w;
        }
    }
}

// ---- the head: out = conv3x3( SiLU( GroupNorm(x) ) ) with <= 4 output channels, fp32 arithmetic -----------------------------------------------
// One workgroup = one TH x TW tile of output pixels of one frame (one pixel per thread).  The channels are walked in chunks of CC: the chunk's
// normalised + activated inputs of the tile and its one-pixel halo are staged in LDS as fp32 (zero outside the frame: the convolution pads its
// INPUT, i.e. the activated tensor), the chunk's weights as [tap][c][4] so that one wave-uniform 16-byte LDS read feeds 4 FMAs.
constexpr int HEAD_TH = 8, HEAD_TW = 32, HEAD_CC = 32;
template <class E, bool IN32>
__global__ __launch_bounds__(HEAD_TH * HEAD_TW) void head_gn_silu_conv3x3_kernel(
    const void* __restrict__ X, int64_t ldx, int H, int W, int channels, int groups, int frames_per_stat, const float* __restrict__ stats,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ Wt /* [9][C][4] */, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, int cout) {
    constexpr int TH = HEAD_TH, TW = HEAD_TW, CC = HEAD_CC, HW_ = (TH + 2) * (TW + 2), SROW = CC + 1;
    extern __shared__ float sm[];
    float* sa = sm;                              // [HW_][SROW] activated inputs of the chunk
    float* sw = sa + HW_ * SROW;                 // [9][CC][4]
    float* sca = sw + 9 * CC * 4;                // [channels] folded scale
    float* scb = sca + channels;                 // [channels] folded shift
    const int tid = threadIdx.x;
    const int tiles_x = (W + TW - 1) / TW;
    const int f = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int cpg = channels / groups;
    const int sb = f / frames_per_stat;
    for (int c = tid; c < channels; c += TH * TW) {
        const int g = c / cpg;
        const float mean = stats[((int64_t)sb * groups + g) * 2], rstd = stats[((int64_t)sb * groups + g) * 2 + 1];
        const float ga = gamma[c] * rstd;
        sca[c] = ga; scb[c] = beta[c] - mean * ga;
    }
    const int py = tid / TW, px = tid % TW;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t frame_row0 = (int64_t)f * H * W;
    for (int c0 = 0; c0 < channels; c0 += CC) {
        __syncthreads();                         // previous chunk consumed (first pass: sca / scb written)
        // stage: HW_ halo pixels x CC/8 octets
        for (int i = tid; i < HW_ * (CC / 8); i += TH * TW) {
            const int hp = i / (CC / 8), o = i % (CC / 8);
            const int hy = hp / (TW + 2), hx = hp % (TW + 2);
            const int y = ty0 + hy - 1, x = tx0 + hx - 1;
            float v[8];
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                const int64_t row = frame_row0 + (int64_t)y * W + x;
                if constexpr (IN32) {
                    const float* xp = (const float*)X + row * ldx + c0 + o * 8;
                    const float4 a = *(const float4*)xp, b = *(const float4*)(xp + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
                    const uint4 u = *(const uint4*)((const svd_bf16*)X + row * ldx + c0 + o * 8);
                    v[0] = E::lo(u.x); v[1] = E::hi(u.x); v[2] = E::lo(u.y); v[3] = E::hi(u.y);
                    v[4] = E::lo(u.z); v[5] = E::hi(u.z); v[6] = E::lo(u.w); v[7] = E::hi(u.w);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float t = v[k] * sca[c0 + o * 8 + k] + scb[c0 + o * 8 + k];
                    v[k] = t / (1.0f + __expf(-t));
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sa[hp * SROW + o * 8 + k] = v[k];
        }
        for (int i = tid; i < 9 * CC; i += TH * TW) {
            const int tap = i / CC, c = i % CC;
            *(float4*)(sw + i * 4) = *(const float4*)(Wt + ((int64_t)tap * channels + c0 + c) * 4);
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const float* ap = sa + ((py + dy) * (TW + 2) + px + dx) * SROW;
            const float* wp = sw + tap * CC * 4;
#pragma unroll 8
            for (int c = 0; c < CC; ++c) {
                const float a = ap[c];
                const float4 w4 = *(const float4*)(wp + c * 4);
                acc[0] = fmaf(a, w4.x, acc[0]); acc[1] = fmaf(a, w4.y, acc[1]);
                acc[2] = fmaf(a, w4.z, acc[2]); acc[3] = fmaf(a, w4.w, acc[3]);
            }
        }
    }
    const int y = ty0 + py, x = tx0 + px;
    if (y < H && x < W) {
        float* yp = Y + (frame_row0 + (int64_t)y * W + x) * ldy;
        for (int k = 0; k < cout; ++k) yp[k] = acc[k] + bias[k];
    }
}

inline unsigned grid_for(int64_t n, int bs = 256) {
    int64_t g = (n + bs - 1) / bs;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int svd_nchw_to_tokens_x3(const float* X0, int32_t c0, const float* X1, int32_t c1, const float* scale, svd_bf16* Y, int32_t cpad,
                                     int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream) {
    if (!X0 || !Y || c0 <= 0 || c1 < 0 || (c1 > 0 && !X1) || cpad % 8 || cpad < c0 + c1 || frames <= 0 || pix <= 0) return SVD_EINVAL;
    if ((uintptr_t)Y & 15) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nchw_to_tokens_x3_kernel<E>, dim3(grid_for((int64_t)frames * pix)), dim3(256), 0,
                                                 (hipStream_t)stream, X0, c0, X1, c1, scale, Y, cpad, frames, pix));
    SVD_CHECK_LAUNCH("nchw_to_tokens_x3");
    return SVD_OK;
}

extern "C" int svd_rows_split3(const float* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels, const float* gamma,
                               const float* beta, float eps, int32_t flags, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || rows <= 0 || channels <= 0 || channels % 8 || channels > 4096 || ldx % 4 || ldy % 8 || ldy < 3 * channels) return SVD_EINVAL;
    if ((flags & SVD_SPLIT3_LAYERNORM) && (!gamma || !beta || channels > 512)) return SVD_EINVAL;      // row statistics: one octet per lane
    if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return SVD_EINVAL;
    const int octets = channels / 8;
    const int lpr = octets <= 4 ? 4 : octets <= 8 ? 8 : octets <= 16 ? 16 : octets <= 32 ? 32 : 64;
    const int rows_per_block = 4 * (64 / lpr);
    int64_t nb = (rows + rows_per_block - 1) / rows_per_block;
    if (nb > 256 * 32) nb = 256 * 32;
    const int do_ln = (flags & SVD_SPLIT3_LAYERNORM) != 0, do_silu = (flags & SVD_SPLIT3_SILU) != 0;
#define S3_LAUNCH(L)                                                                                                                          \
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((rows_split3_kernel<L, E>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, X, ldx, Y, ldy, \
                                                 rows, channels, gamma, beta, eps, do_ln, do_silu))
    if (lpr == 4) S3_LAUNCH(4);
    else if (lpr == 8) S3_LAUNCH(8);
    else if (lpr == 16) S3_LAUNCH(16);
    else if (lpr == 32) S3_LAUNCH(32);
    else S3_LAUNCH(64);
#undef S3_LAUNCH
    SVD_CHECK_LAUNCH("rows_split3");
    return SVD_OK;
}

extern "C" int svd_add_rows_bf32(const void* X, int64_t ldx, const float* B, int64_t ldb, void* Y, int64_t ldy, int64_t rows, int32_t channels,
                                 int32_t dtype, svd_stream_t stream) {
    if (!X || !B || !Y || channels <= 0 || channels % 8 || ldx % 8 || ldb % 4 || ldy % 8 || rows <= 0) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)B | (uintptr_t)Y) & 15) return SVD_EINVAL;
    if (dtype & SVD_DTYPE_IN_F32) {
        SVD_DISPATCH_DTYPE(dtype & 0xff, hipLaunchKernelGGL((add_rows_bf32_kernel<E, true>), dim3(grid_for(rows * (channels / 8))), dim3(256), 0,
                                                            (hipStream_t)stream, X, ldx, B, ldb, Y, ldy, rows, channels));
    } else {
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((add_rows_bf32_kernel<E, false>), dim3(grid_for(rows * (channels / 8))), dim3(256), 0,
                                                     (hipStream_t)stream, X, ldx, B, ldb, Y, ldy, rows, channels));
    }
    SVD_CHECK_LAUNCH("add_rows_bf32");
    return SVD_OK;
}

extern "C" int svd_head_gn_silu_conv3x3(const void* X, int64_t ldx, int32_t frames, int32_t H, int32_t W, int32_t channels, int32_t groups,
                                        int32_t frames_per_stat, const float* stats, const float* gamma, const float* beta, const float* Wt,
                                        const float* bias, float* Y, int64_t ldy, int32_t cout, int32_t dtype, svd_stream_t stream) {
    if (!X || !stats || !gamma || !beta || !Wt || !bias || !Y || frames <= 0 || H <= 0 || W <= 0) return SVD_EINVAL;
    if (channels <= 0 || channels % HEAD_CC || groups <= 0 || channels % groups || cout < 1 || cout > 4 || ldy < cout || ldx % 8) return SVD_EINVAL;
    if (frames_per_stat <= 0 || frames % frames_per_stat || frames > 65535) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wt) & 15) return SVD_EINVAL;
    const int tiles = ((H + HEAD_TH - 1) / HEAD_TH) * ((W + HEAD_TW - 1) / HEAD_TW);
    const size_t lds = ((size_t)(HEAD_TH + 2) * (HEAD_TW + 2) * (HEAD_CC + 1) + 9 * HEAD_CC * 4 + 2 * (size_t)channels) * sizeof(float);
    if (lds > 64 * 1024) return SVD_EINVAL;
    const int base_dt = dtype & 0xff;
    if (dtype & SVD_DTYPE_IN_F32) {
        SVD_DISPATCH_DTYPE(base_dt, hipLaunchKernelGGL((head_gn_silu_conv3x3_kernel<E, true>), dim3(tiles, frames), dim3(HEAD_TH * HEAD_TW), lds,
                                                       (hipStream_t)stream, X, ldx, H, W, channels, groups, frames_per_stat, stats, gamma, beta, Wt,
                                                       bias, Y, ldy, cout));
    } else {
        SVD_DISPATCH_DTYPE(base_dt, hipLaunchKernelGGL((head_gn_silu_conv3x3_kernel<E, false>), dim3(tiles, frames), dim3(HEAD_TH * HEAD_TW), lds,
                                                       (hipStream_t)stream, X, ldx, H, W, channels, groups, frames_per_stat, stats, gamma, beta, Wt,
                                                       bias, Y, ldy, cout));
    }
    SVD_CHECK_LAUNCH("head_gn_silu_conv3x3");
    return SVD_OK;
}
