// EMA-VFI frame interpolation (reference: code/i2v_enhance/thirdparty/VFI): the pieces that are not GEMM/conv/LayerNorm-shaped.
// All HBM-bound gathers / elementwise passes except the 7x7-window inter-frame attention, which is ~1 GFLOP per frame pair at
// 720x1280 and runs as scalar fp32 FMAs out of LDS (49 tokens x head_dim 32 per (window, head): too small for an MFMA tiling to pay).
#include "svd_common.h"

namespace {

constexpr int WIN_TOK = 49;   // 7 x 7 window (config.py:10 W = 7)
constexpr int WIN_HD = 32;    // head dim: dim / heads = 8F / (8F / 32)  (config.py:13)
constexpr int WIN_MD_MAX = 16;

// nn.PReLU(C) in place on 16-bit rows: y = x >= 0 ? x : slope[c] * x
template <class E>
__global__ __launch_bounds__(256) void prelu_rows_kernel(svd_bf16* __restrict__ X, int64_t ldx, int64_t rows, int octets,
                                                        const float* __restrict__ slope) {
    const int64_t total = rows * octets;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / octets; const int o = (int)(i - r * octets);
        uint4* px = (uint4*)(X + r * ldx + o * 8);
        const uint4 u = *px;
        const float4 s0 = *(const float4*)(slope + o * 8), s1 = *(const float4*)(slope + o * 8 + 4);
        float v[8] = {E::lo(u.x), E::hi(u.x), E::lo(u.y), E::hi(u.y), E::lo(u.z), E::hi(u.z), E::lo(u.w), E::hi(u.w)};
        const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] >= 0.f ? v[k] : v[k] * s[k];
        uint4 w;
        w.x = E::pack(v[0], v[1]); w.y = E::pack(v[2], v[3]); w.z = E::pack(v[4], v[5]); w.w = E::pack(v[6], v[7]);
        *px = w;
    }
}

__global__ __launch_bounds__(256) void prelu_rows_f32_kernel(float* __restrict__ X, int64_t ldx, int64_t rows, int C, const float* __restrict__ slope) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C; const int c = (int)(i - r * C);
        const float v = X[r * ldx + c];
        X[r * ldx + c] = v >= 0.f ? v : v * slope[c];
    }
}

// Mlp.dwconv + act (feature_extractor.py:104-108, 505-515): depthwise 3x3, padding 1, bias, exact GELU; channels-last rows.
// W9: [9][C] fp32 (tap-major so that the 8 channels of a thread are contiguous).
template <class E>
__global__ __launch_bounds__(256) void dwconv3x3_gelu_kernel(const svd_bf16* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y,
                                                            int64_t ldy, const float* __restrict__ W9, const float* __restrict__ bias,
                                                            int frames, int H, int W, int C) {
    const int octets = C >> 3;
    const int64_t total = (int64_t)frames * H * W * octets;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pixel = i / octets; const int o = (int)(i - pixel * octets);
        const int x = (int)(pixel % W); const int64_t t = pixel / W; const int y = (int)(t % H);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = bias[o * 8 + k];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const uint4 u = *(const uint4*)(X + (pixel + (int64_t)dy * W + dx) * ldx + o * 8);
                const float* w = W9 + (int64_t)((dy + 1) * 3 + (dx + 1)) * C + o * 8;
                const float4 w0 = *(const float4*)w, w1 = *(const float4*)(w + 4);
                acc[0] = fmaf(E::lo(u.x), w0.x, acc[0]); acc[1] = fmaf(E::hi(u.x), w0.y, acc[1]);
                acc[2] = fmaf(E::lo(u.y), w0.z, acc[2]); acc[3] = fmaf(E::hi(u.y), w0.w, acc[3]);
                acc[4] = fmaf(E::lo(u.z), w1.x, acc[4]); acc[5] = fmaf(E::hi(u.z), w1.y, acc[5]);
                acc[6] = fmaf(E::lo(u.w), w1.z, acc[6]); acc[7] = fmaf(E::hi(u.w), w1.w, acc[7]);
            }
        }
        uint4 w;
        w.x = E::pack(gelu_erf_f(acc[0]), gelu_erf_f(acc[1])); w.y = E::pack(gelu_erf_f(acc[2]), gelu_erf_f(acc[3]));
        w.z = E::pack(gelu_erf_f(acc[4]), gelu_erf_f(acc[5])); w.w = E::pack(gelu_erf_f(acc[6]), gelu_erf_f(acc[7]));
        *(uint4*)(Y + pixel * ldy + o * 8) = w;
    }
}

// InterFrameAttention.forward (feature_extractor.py:141-171) for one (window, head) per 64-lane workgroup.
//   Q [n_win*49][ldq]: q of window w;  KV [n_win*49][ldkv]: (k | v) of every window -- window w attends to window (w + n_win/2) % n_win,
//   i.e. the same window of the OTHER frame (x_reverse, :264);  CE [n_win*49][ldce] fp32: cor_embed of window w (md = heads * mdh).
//   mask [n_mask][49][49] fp32 or null, window w uses mask[w % n_mask] (:155-160).
//   OX [n_win*49][ldo] = softmax(q k^T * scale + mask) v ;  OC [n_win*49][ldc] = softmax(..) ce - ce   (c_reverse - cor_embed_, :168).
template <class E>
__global__ __launch_bounds__(64) void window_attn_kernel(const svd_bf16* __restrict__ Q, int64_t ldq, const svd_bf16* __restrict__ KV,
                                                        int64_t ldkv, const float* __restrict__ CE, int64_t ldce,
                                                        const float* __restrict__ mask, int n_mask, svd_bf16* __restrict__ OX, int64_t ldo,
                                                        svd_bf16* __restrict__ OC, int64_t ldc, int n_win, int heads, int mdh, float scale) {
    __shared__ float sk[WIN_TOK][WIN_HD + 1];
    __shared__ float sv[WIN_TOK][WIN_HD + 1];
    __shared__ float sc[WIN_TOK][WIN_MD_MAX + 1];
    const int w = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int wo = (w + n_win / 2) % n_win;
    const int C = heads * WIN_HD;
    for (int i = lane; i < WIN_TOK * WIN_HD; i += 64) {
        const int j = i / WIN_HD, d = i - j * WIN_HD;
        const svd_bf16* row = KV + ((int64_t)wo * WIN_TOK + j) * ldkv + h * WIN_HD + d;
        sk[j][d] = E::to_f32(row[0]);
        sv[j][d] = E::to_f32(row[C]);
    }
    for (int i = lane; i < WIN_TOK * mdh; i += 64) {
        const int j = i / mdh, d = i - j * mdh;
        sc[j][d] = CE[((int64_t)w * WIN_TOK + j) * ldce + h * mdh + d];
    }
    __syncthreads();
    if (lane >= WIN_TOK) return;
    const int64_t r = (int64_t)w * WIN_TOK + lane;
    float q[WIN_HD];
#pragma unroll
    for (int d = 0; d < WIN_HD; ++d) q[d] = E::to_f32(Q[r * ldq + h * WIN_HD + d]);
    const float* mrow = mask ? mask + ((int64_t)(w % n_mask) * WIN_TOK + lane) * WIN_TOK : nullptr;
    float s[WIN_TOK];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < WIN_TOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < WIN_HD; ++d) a = fmaf(q[d], sk[j][d], a);
        a *= scale;
        if (mrow) a += mrow[j];
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < WIN_TOK; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    const float inv = 1.0f / sum;
    float o[WIN_HD];
#pragma unroll
    for (int d = 0; d < WIN_HD; ++d) o[d] = 0.f;
    float oc[WIN_MD_MAX];
#pragma unroll
    for (int d = 0; d < WIN_MD_MAX; ++d) oc[d] = 0.f;
#pragma unroll
    for (int j = 0; j < WIN_TOK; ++j) {
        const float p = s[j] * inv;
#pragma unroll
        for (int d = 0; d < WIN_HD; ++d) o[d] = fmaf(p, sv[j][d], o[d]);
#pragma unroll
        for (int d = 0; d < WIN_MD_MAX; ++d) if (d < mdh) oc[d] = fmaf(p, sc[j][d], oc[d]);
    }
#pragma unroll
    for (int d = 0; d < WIN_HD; ++d) OX[r * ldo + h * WIN_HD + d] = E::from_f32(o[d]);
#pragma unroll
    for (int d = 0; d < WIN_MD_MAX; ++d) if (d < mdh) OC[r * ldc + h * mdh + d] = E::from_f32(oc[d] - sc[lane][d]);
}

// warp (warplayer.py:7-22): backward warp by a pixel-unit flow = grid_sample(bilinear, padding_mode='border', align_corners=True) at
// (x + flow_x, y + flow_y).  Channels-last source/destination, VEC channels per thread; flow: 2 fp32 per pixel at FLOW[pixel * ldf].
template <class E, bool F32>
__global__ __launch_bounds__(256) void warp_bilinear_kernel(const void* __restrict__ Xv, int64_t ldx, void* __restrict__ Yv, int64_t ldy,
                                                           const float* __restrict__ FLOW, int64_t ldf, int frames, int H, int W, int C) {
    constexpr int VEC = F32 ? 1 : 8;
    const int groups = C / VEC;
    const int64_t total = (int64_t)frames * H * W * groups;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pixel = i / groups; const int g = (int)(i - pixel * groups);
        const int x = (int)(pixel % W); const int64_t t = pixel / W; const int y = (int)(t % H); const int64_t f = t / H;
        float fx = (float)x + FLOW[pixel * ldf], fy = (float)y + FLOW[pixel * ldf + 1];
        fx = fminf(fmaxf(fx, 0.f), (float)(W - 1)); fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
        const int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
        const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
        const float ax = fx - (float)x0, ay = fy - (float)y0;
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        const int64_t base = f * H * W;
        const int64_t p00 = base + (int64_t)y0 * W + x0, p01 = base + (int64_t)y0 * W + x1, p10 = base + (int64_t)y1 * W + x0,
                      p11 = base + (int64_t)y1 * W + x1;
        if constexpr (F32) {
            const float* X = (const float*)Xv;
            ((float*)Yv)[pixel * ldy + g] = X[p00 * ldx + g] * w00 + X[p01 * ldx + g] * w01 + X[p10 * ldx + g] * w10 + X[p11 * ldx + g] * w11;
        } else {
            const svd_bf16* X = (const svd_bf16*)Xv;
            const uint4 a = *(const uint4*)(X + p00 * ldx + g * 8), b = *(const uint4*)(X + p01 * ldx + g * 8),
                        c = *(const uint4*)(X + p10 * ldx + g * 8), d = *(const uint4*)(X + p11 * ldx + g * 8);
            const uint32_t av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w}, dv[4] = {d.x, d.y, d.z, d.w};
            uint32_t r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float lo = E::lo(av[k]) * w00 + E::lo(bv[k]) * w01 + E::lo(cv[k]) * w10 + E::lo(dv[k]) * w11;
                const float hi = E::hi(av[k]) * w00 + E::hi(bv[k]) * w01 + E::hi(cv[k]) * w10 + E::hi(dv[k]) * w11;
                r[k] = E::pack(lo, hi);
            }
            *(uint4*)((svd_bf16*)Yv + pixel * ldy + g * 8) = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
}

// F.interpolate(mode='bilinear', align_corners=False, scale_factor=s) on channels-last fp32 [frames][hin][win][ldx], C <= ldx channels:
// src = (dst + 0.5) * (1 / s) - 0.5 clamped at 0 (PyTorch's area_pixel_compute_source_index with the given scale), then
// out[c] = (accumulate ? out[c] : 0) + mult[c] * value   (the "* scale" factors of flow_estimation.py:32,38-39,64 and "flow + flow_d" :129).
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ X, int64_t ldx, float* __restrict__ Y, int64_t ldy,
                                                             int frames, int hin, int win, int hout, int wout, int C, float rh, float rw,
                                                             const float* __restrict__ mult, int accumulate) {
    const int64_t total = (int64_t)frames * hout * wout * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C); const int64_t pixel = i / C;
        const int x = (int)(pixel % wout); const int64_t t = pixel / wout; const int y = (int)(t % hout); const int64_t f = t / hout;
        const float sy = fmaxf(((float)y + 0.5f) * rh - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * rw - 0.5f, 0.f);
        const int y0 = min((int)sy, hin - 1), x0 = min((int)sx, win - 1);
        const int y1 = y0 + (y0 < hin - 1 ? 1 : 0), x1 = x0 + (x0 < win - 1 ? 1 : 0);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float* src = X + f * hin * win * ldx + c;
        const float v = (1.f - ly) * ((1.f - lx) * src[((int64_t)y0 * win + x0) * ldx] + lx * src[((int64_t)y0 * win + x1) * ldx]) +
                        ly * ((1.f - lx) * src[((int64_t)y1 * win + x0) * ldx] + lx * src[((int64_t)y1 * win + x1) * ldx]);
        const float m = mult ? mult[c] : 1.f;
        float* dst = Y + pixel * ldy + c;
        *dst = (accumulate ? *dst : 0.f) + m * v;
    }
}

// MultiScaleFlow.forward :131-139: merged = w0 * sigmoid(mask) + w1 * (1 - sigmoid(mask)); res = sigmoid(u[:3]) * 2 - 1 (refine.py:71);
// pred = clamp(merged + res, 0, 1).  W0/W1 [n][3], MASK at MASKP[pixel * ldm], U [n][ldu] (the Unet's last conv, pre-sigmoid), fp32.
__global__ __launch_bounds__(256) void vfi_merge_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ MASKP,
                                                       int64_t ldm, const float* __restrict__ U, int64_t ldu, float* __restrict__ merged,
                                                       float* __restrict__ pred, int64_t npix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix * 3; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / 3; const int c = (int)(i - p * 3);
        const float sg = 1.f / (1.f + __expf(-MASKP[p * ldm]));
        const float m = W0[i] * sg + W1[i] * (1.f - sg);
        const float res = 2.f / (1.f + __expf(-U[p * ldu + c])) - 1.f;
        if (merged) merged[i] = m;
        pred[i] = fminf(fmaxf(m + res, 0.f), 1.f);
    }
}

// Trainer.Model.inference fast_TTA (:90-94): (pred[0] + rot180(pred[1])) / 2 on channels-last [2][H][W][3]; optional uint8 copy with the
// truncation of vfi_process (i2v_enhance_interface.py:46-47: (x * 255.0).astype(np.uint8)).
__global__ __launch_bounds__(256) void vfi_tta_average_kernel(const float* __restrict__ P, float* __restrict__ out, uint8_t* __restrict__ out8,
                                                             int H, int W) {
    const int64_t npix = (int64_t)H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix * 3; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i / 3; const int c = (int)(i - p * 3);
        const float v = __fdiv_rn(__fadd_rn(P[i], P[(npix + (npix - 1 - p)) * 3 + c]), 2.0f);
        if (out) out[i] = v;
        if (out8) out8[i] = (uint8_t)(int)__fmul_rn(v, 255.0f);
    }
}

inline unsigned blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

}  // namespace

extern "C" int svd_prelu_rows(void* X, int64_t ldx, int64_t rows, int32_t channels, const float* slope, int32_t dtype, svd_stream_t stream) {
    if (dtype == SVD_DTYPE_F32) {
        if (!X || !slope || rows <= 0 || channels <= 0 || ldx < channels) return SVD_EINVAL;
        hipLaunchKernelGGL(prelu_rows_f32_kernel, dim3(blocks_for(rows * channels)), dim3(256), 0, (hipStream_t)stream, (float*)X, ldx, rows, channels,
                           slope);
        SVD_CHECK_LAUNCH("prelu_rows");
        return SVD_OK;
    }
    if (!X || !slope || rows <= 0 || channels <= 0 || channels % 8 || ldx % 8 || ((uintptr_t)X & 15) || ((uintptr_t)slope & 15)) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(prelu_rows_kernel<E>, dim3(blocks_for(rows * (channels / 8))), dim3(256), 0, (hipStream_t)stream,
                                                 (svd_bf16*)X, ldx, rows, channels / 8, slope));
    SVD_CHECK_LAUNCH("prelu_rows");
    return SVD_OK;
}

extern "C" int svd_dwconv3x3_gelu(const svd_bf16* X, int64_t ldx, svd_bf16* Y, int64_t ldy, const float* w9, const float* bias, int32_t frames,
                                  int32_t h, int32_t w, int32_t channels, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || !w9 || !bias || frames <= 0 || h <= 0 || w <= 0 || channels <= 0 || channels % 8 || ldx % 8 || ldy % 8) return SVD_EINVAL;
    if ((((uintptr_t)X | (uintptr_t)Y | (uintptr_t)w9) & 15) || X == Y) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(dwconv3x3_gelu_kernel<E>, dim3(blocks_for((int64_t)frames * h * w * (channels / 8))), dim3(256), 0,
                                                 (hipStream_t)stream, X, ldx, Y, ldy, w9, bias, frames, h, w, channels));
    SVD_CHECK_LAUNCH("dwconv3x3_gelu");
    return SVD_OK;
}

extern "C" int svd_window_attn_7x7(const svd_bf16* Q, int64_t ldq, const svd_bf16* KV, int64_t ldkv, const float* CE, int64_t ldce,
                                   const float* mask, int32_t n_mask, svd_bf16* OX, int64_t ldo, svd_bf16* OC, int64_t ldc, int32_t n_win,
                                   int32_t heads, int32_t motion_per_head, float scale, int32_t dtype, svd_stream_t stream) {
    if (!Q || !KV || !CE || !OX || !OC || n_win <= 0 || n_win % 2 || heads <= 0 || motion_per_head <= 0 || motion_per_head > WIN_MD_MAX)
        return SVD_EINVAL;
    if (mask && (n_mask <= 0 || n_win % n_mask)) return SVD_EINVAL;
    if (ldq < heads * WIN_HD || ldkv < 2 * heads * WIN_HD || ldo < heads * WIN_HD || ldce < heads * motion_per_head || ldc < heads * motion_per_head)
        return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(window_attn_kernel<E>, dim3(n_win, heads), dim3(64), 0, (hipStream_t)stream, Q, ldq, KV, ldkv, CE,
                                                 ldce, mask, n_mask, OX, ldo, OC, ldc, n_win, heads, motion_per_head, scale));
    SVD_CHECK_LAUNCH("window_attn_7x7");
    return SVD_OK;
}

extern "C" int svd_warp_bilinear(const void* X, int64_t ldx, void* Y, int64_t ldy, const float* flow, int64_t ldf, int32_t frames, int32_t h,
                                 int32_t w, int32_t channels, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || !flow || X == Y || frames <= 0 || h <= 0 || w <= 0 || channels <= 0 || ldx < channels || ldy < channels || ldf < 2)
        return SVD_EINVAL;
    const unsigned nb_f32 = blocks_for((int64_t)frames * h * w * channels);
    if (dtype == SVD_DTYPE_F32) {
        hipLaunchKernelGGL((warp_bilinear_kernel<ElemBF16, true>), dim3(nb_f32), dim3(256), 0, (hipStream_t)stream, X, ldx, Y, ldy, flow, ldf,
                           frames, h, w, channels);
    } else {
        if (channels % 8 || ldx % 8 || ldy % 8 || (((uintptr_t)X | (uintptr_t)Y) & 15)) return SVD_EINVAL;
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((warp_bilinear_kernel<E, false>), dim3(blocks_for((int64_t)frames * h * w * (channels / 8))),
                                                     dim3(256), 0, (hipStream_t)stream, X, ldx, Y, ldy, flow, ldf, frames, h, w, channels));
    }
    SVD_CHECK_LAUNCH("warp_bilinear");
    return SVD_OK;
}

extern "C" int svd_resize_bilinear_f32(const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t frames, int32_t hin, int32_t win, int32_t hout,
                                       int32_t wout, int32_t channels, float scale_factor, const float* mult, int32_t accumulate,
                                       svd_stream_t stream) {
    if (!X || !Y || X == Y || frames <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || channels <= 0 || ldx < channels || ldy < channels ||
        !(scale_factor > 0.f))
        return SVD_EINVAL;
    const float r = 1.0f / scale_factor;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for((int64_t)frames * hout * wout * channels)), dim3(256), 0, (hipStream_t)stream, X, ldx,
                       Y, ldy, frames, hin, win, hout, wout, channels, r, r, mult, accumulate);
    SVD_CHECK_LAUNCH("resize_bilinear_f32");
    return SVD_OK;
}

extern "C" int svd_vfi_merge(const float* warped0, const float* warped1, const float* mask, int64_t ld_mask, const float* unet_out, int64_t ld_unet,
                             float* merged, float* pred, int64_t n_pixels, svd_stream_t stream) {
    if (!warped0 || !warped1 || !mask || !unet_out || !pred || n_pixels <= 0 || ld_mask < 1 || ld_unet < 3) return SVD_EINVAL;
    hipLaunchKernelGGL(vfi_merge_kernel, dim3(blocks_for(n_pixels * 3)), dim3(256), 0, (hipStream_t)stream, warped0, warped1, mask, ld_mask, unet_out,
                       ld_unet, merged, pred, n_pixels);
    SVD_CHECK_LAUNCH("vfi_merge");
    return SVD_OK;
}

extern "C" int svd_vfi_tta_average(const float* pred2, float* out, uint8_t* out_u8, int32_t h, int32_t w, svd_stream_t stream) {
    if (!pred2 || (!out && !out_u8) || h <= 0 || w <= 0) return SVD_EINVAL;
    hipLaunchKernelGGL(vfi_tta_average_kernel, dim3(blocks_for((int64_t)h * w * 3)), dim3(256), 0, (hipStream_t)stream, pred2, out, out_u8, h, w);
    SVD_CHECK_LAUNCH("vfi_tta_average");
    return SVD_OK;
}
