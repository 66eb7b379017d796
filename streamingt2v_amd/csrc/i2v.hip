// Kernels that only the I2VGen-XL enhancement stage needs.  gfx950 only.
#include "svd_common.h"

namespace {

// nn.AdaptiveAvgPool2d((hout, wout)) on channels-last tokens (image_latents_context_embedding[2], unet_i2vgen_xl.py:265):
// window of output (oy, ox) = rows [floor(oy*hin/hout), ceil((oy+1)*hin/hout)) x cols likewise.  One thread per 8 channels.
template <class E>
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const svd_bf16* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y,
                                                               int64_t ldy, int frames, int hin, int win, int hout, int wout, int c8) {
    const int64_t total = (int64_t)frames * hout * wout * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)(i % c8);
        int64_t r = i / c8;
        const int ox = (int)(r % wout); r /= wout;
        const int oy = (int)(r % hout);
        const int f = (int)(r / hout);
        const int y0 = (oy * hin) / hout, y1 = ((oy + 1) * hin + hout - 1) / hout;
        const int x0 = (ox * win) / wout, x1 = ((ox + 1) * win + wout - 1) / wout;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const uint4 u = *(const uint4*)(X + ((int64_t)(f * hin + y) * win + x) * ldx + cc * 8);
                acc[0] += E::lo(u.x); acc[1] += E::hi(u.x); acc[2] += E::lo(u.y); acc[3] += E::hi(u.y);
                acc[4] += E::lo(u.z); acc[5] += E::hi(u.z); acc[6] += E::lo(u.w); acc[7] += E::hi(u.w);
            }
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        uint4 o;
        o.x = E::pack(acc[0] * inv, acc[1] * inv); o.y = E::pack(acc[2] * inv, acc[3] * inv);
        o.z = E::pack(acc[4] * inv, acc[5] * inv); o.w = E::pack(acc[6] * inv, acc[7] * inv);
        *(uint4*)(Y + ((int64_t)(f * hout + oy) * wout + ox) * ldy + cc * 8) = o;
    }
}

// I2VGenXLTransformerTemporalEncoder on the 4-channel image latents (unet_i2vgen_xl.py:110-160, called at :700-709): per pixel a
// sequence of F frames x 4 channels:  x += to_out(attn(LN(x)))  (2 heads x 4, q/k/v without bias) ;  x += W2 gelu(W1 x + b1) + b2.
// One wave per (batch, pixel); a lane owns frames `lane` and `lane + 64` (F <= 128: the 38-frame windows use one slot, the
// single 100-frame window of the no-blending mode two); keys/values travel by cross-lane reads.  Everything fp32.
// params: ln_w[4] ln_b[4] wq[8][4] wk[8][4] wv[8][4] wo[4][8] bo[4] w1[16][4] b1[16] w2[4][16] b2[4]   (288 floats)
template <class E, bool IN32>
__global__ __launch_bounds__(256) void i2v_image_encoder_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const float* __restrict__ prm,
                                                                float* __restrict__ Y, int batch, int frames, int pix) {
    __shared__ float P[288];
    for (int i = threadIdx.x; i < 288; i += 256) P[i] = prm[i];
    __syncthreads();
    const float *ln_w = P, *ln_b = P + 4, *wq = P + 8, *wk = P + 40, *wv = P + 72, *wo = P + 104, *bo = P + 136, *w1 = P + 140,
                *b1 = P + 204, *w2 = P + 220, *b2 = P + 284;
    const int lane = threadIdx.x & 63;
    const int nslot = (frames + 63) >> 6;                       // 1 or 2 (wave-uniform)
    const int64_t nprob = (int64_t)batch * pix;
    for (int64_t prob = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); prob < nprob; prob += (int64_t)gridDim.x * 4) {
        const int b = (int)(prob / pix), p = (int)(prob % pix);
        float x[2][4], q[2][8], k[2][8], v[2][8];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const int fr0 = lane + 64 * sl;
            const int fr = fr0 < frames ? fr0 : frames - 1;
            const int64_t row = ((int64_t)b * frames + fr) * pix + p;
            if constexpr (IN32) {              // fp32 rows (the enhancer's precision plan: the image-latent projection stays fp32 up to here)
                const float4 u = *(const float4*)((const float*)X + row * ldx);
                x[sl][0] = u.x; x[sl][1] = u.y; x[sl][2] = u.z; x[sl][3] = u.w;
            } else {
                const uint2 u = *(const uint2*)(X + row * ldx);
                x[sl][0] = E::lo(u.x); x[sl][1] = E::hi(u.x); x[sl][2] = E::lo(u.y); x[sl][3] = E::hi(u.y);
            }
            // LayerNorm(4), eps 1e-5
            const float mean = 0.25f * ((x[sl][0] + x[sl][1]) + (x[sl][2] + x[sl][3]));
            float var = 0.f, n[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { n[i] = x[sl][i] - mean; var += n[i] * n[i]; }
            const float rstd = rsqrtf(0.25f * var + 1e-5f);
#pragma unroll
            for (int i = 0; i < 4; ++i) n[i] = n[i] * rstd * ln_w[i] + ln_b[i];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                q[sl][o] = k[sl][o] = v[sl][o] = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    q[sl][o] += wq[o * 4 + i] * n[i]; k[sl][o] += wk[o * 4 + i] * n[i]; v[sl][o] += wv[o * 4 + i] * n[i];
                }
            }
        }
        // attention over the frames: two heads of dim 4, scale 4^-0.5; two passes (max, then sum) keep it simple and exact
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (sl >= nslot) break;
            float att[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float mx = -INFINITY;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int nj = frames - 64 * ks < 64 ? frames - 64 * ks : 64;
                    for (int j = 0; j < nj; ++j) {
                        float s = 0.f;
#pragma unroll
                        for (int d = 0; d < 4; ++d) s += q[sl][h * 4 + d] * __shfl(k[ks][h * 4 + d], j, 64);
                        mx = fmaxf(mx, s * 0.5f);
                    }
                }
                float l = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int nj = frames - 64 * ks < 64 ? frames - 64 * ks : 64;
                    for (int j = 0; j < nj; ++j) {
                        float s = 0.f;
#pragma unroll
                        for (int d = 0; d < 4; ++d) s += q[sl][h * 4 + d] * __shfl(k[ks][h * 4 + d], j, 64);
                        const float pj = __expf(s * 0.5f - mx);
                        l += pj;
#pragma unroll
                        for (int d = 0; d < 4; ++d) o[d] += pj * __shfl(v[ks][h * 4 + d], j, 64);
                    }
                }
                const float inv = 1.f / l;
#pragma unroll
                for (int d = 0; d < 4; ++d) att[h * 4 + d] = o[d] * inv;
            }
            float xo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = bo[i];
#pragma unroll
                for (int o = 0; o < 8; ++o) a += wo[i * 8 + o] * att[o];
                xo[i] = x[sl][i] + a;
            }
            float hdn[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                float a = b1[o];
#pragma unroll
                for (int i = 0; i < 4; ++i) a += w1[o * 4 + i] * xo[i];
                hdn[o] = 0.5f * a * (1.f + erff(a * 0.70710678118654752f));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = b2[i];
#pragma unroll
                for (int o = 0; o < 16; ++o) a += w2[i * 16 + o] * hdn[o];
                xo[i] += a;
            }
            const int fr = lane + 64 * sl;
            if (fr < frames) {  // fp32 NCHW [(b f), 4, pix]: the per-step sample is concatenated to it by svd_nchw_to_tokens
                float* y = Y + ((int64_t)b * frames + fr) * 4 * pix + p;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[(int64_t)i * pix] = xo[i];
            }
        }
    }
}

// One DDIM step (eta 0) with classifier-free guidance on fp32 latents, layout [(b f), C, pix] NCHW per frame:
//   v = vu + g (vc - vu);  v-prediction: x0 = sa x - sb v, eps = sa v + sb x;  epsilon: x0 = (x - sb v) / sa, eps = v
//   x_prev = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps            (diffusers DDIMScheduler.step; pipeline_i2vgen_xl.py:872-885)
__global__ __launch_bounds__(256) void ddim_cfg_step_kernel(const float* __restrict__ x, const float* __restrict__ vu, const float* __restrict__ vc,
                                                            float* __restrict__ out, int64_t n, float g, float sa, float sb, float spa,
                                                            float spb, int vpred) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = vu[i];
        const float v = vc ? u + g * (vc[i] - u) : u;
        const float xi = x[i];
        float x0, eps;
        if (vpred) { x0 = sa * xi - sb * v; eps = sa * v + sb * xi; }
        else { x0 = (xi - sb * v) / sa; eps = v; }
        out[i] = spa * x0 + spb * eps;
    }
}

}  // namespace

extern "C" int svd_adaptive_avgpool_tokens(const svd_bf16* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t hin,
                                           int32_t win, int32_t hout, int32_t wout, int32_t channels, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || frames <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0 || channels <= 0) return SVD_EINVAL;
    if (channels % 8 || ldx % 8 || ldy % 8 || (((uintptr_t)X | (uintptr_t)Y) & 15)) return SVD_EINVAL;
    const int64_t total = (int64_t)frames * hout * wout * (channels / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(adaptive_avgpool_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, ldx,
                                                 Y, ldy, frames, hin, win, hout, wout, channels / 8));
    SVD_CHECK_LAUNCH("adaptive_avgpool_tokens");
    return SVD_OK;
}

extern "C" int svd_i2v_image_temporal_encoder(const svd_bf16* X, int64_t ldx, const float* params, float* Y, int32_t batch,
                                              int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream) {
    if (!X || !params || !Y || batch <= 0 || frames <= 0 || frames > 128 || pix <= 0) return SVD_EINVAL;
    const bool in32 = (dtype & SVD_DTYPE_IN_F32) != 0;
    if (ldx % 4 || ((uintptr_t)X & (in32 ? 15 : 7))) return SVD_EINVAL;
    int64_t blocks = ((int64_t)batch * pix + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    if (in32) {
        SVD_DISPATCH_DTYPE(dtype & 0xff, hipLaunchKernelGGL((i2v_image_encoder_kernel<E, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                                            X, ldx, params, Y, batch, frames, pix));
    } else {
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((i2v_image_encoder_kernel<E, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                                     X, ldx, params, Y, batch, frames, pix));
    }
    SVD_CHECK_LAUNCH("i2v_image_temporal_encoder");
    return SVD_OK;
}

extern "C" int svd_ddim_cfg_step(const float* x, const float* pred_uncond, const float* pred_cond, float* out, int64_t n,
                                 float guidance_scale, float alpha_t, float alpha_prev, int32_t v_prediction, svd_stream_t stream) {
    if (!x || !pred_uncond || !out || n <= 0 || alpha_t <= 0.f || alpha_t > 1.f || alpha_prev < 0.f || alpha_prev > 1.f) return SVD_EINVAL;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ddim_cfg_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, pred_uncond, pred_cond, out, n,
                       guidance_scale, sqrtf(alpha_t), sqrtf(1.f - alpha_t), sqrtf(alpha_prev), sqrtf(1.f - alpha_prev), v_prediction);
    SVD_CHECK_LAUNCH("ddim_cfg_step");
    return SVD_OK;
}
