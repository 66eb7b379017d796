// Layout conversion and small fused elementwise kernels (HBM-bound glue around the MFMA kernels).
#include "svd_common.h"

namespace {

// NCHW fp32 (X0 | X1 on the channel axis) -> channels-last bf16 rows of cpad channels, scaled per frame.
template <class E>
__global__ void nchw_to_tokens_kernel(const float* __restrict__ X0, int c0, const float* __restrict__ X1, int c1,
                                      const float* __restrict__ scale, svd_bf16* __restrict__ Y, int cpad, int frames, int pix) {
    const int64_t total = (int64_t)frames * pix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / pix), p = (int)(i - (int64_t)f * pix);
        const float sc = scale ? scale[f] : 1.f;
        svd_bf16* y = Y + i * cpad;
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
            uint4 w;
            w.x = E::pack(v[0], v[1]); w.y = E::pack(v[2], v[3]);
            w.z = E::pack(v[4], v[5]); w.w = E::pack(v[6], v[7]);
            *(uint4*)(y + cb) = w;
        }
    }
}

template <class E>
__global__ void tokens_to_nchw_kernel(const void* __restrict__ X, int is_f32, int64_t ldx, float* __restrict__ Y, int c,
                                      int frames, int pix) {
    const int64_t total = (int64_t)frames * pix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / pix), p = (int)(i - (int64_t)f * pix);
        for (int k = 0; k < c; ++k) {
            const float v = is_f32 ? ((const float*)X)[i * ldx + k] : E::to_f32(((const svd_bf16*)X)[i * ldx + k]);
            Y[((int64_t)f * c + k) * pix + p] = v;
        }
    }
}

// generic strided row copy of 16-byte vectors: Y[m][off..off+c) = A[m][0..c)
__global__ void copy_rows_kernel(const svd_bf16* __restrict__ A, int64_t lda, int c, svd_bf16* __restrict__ Y, int64_t ldy,
                                 int64_t rows) {
    const int oct = c >> 3;
    const int64_t total = rows * oct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / oct;
        const int o = (int)(i - r * oct);
        *(uint4*)(Y + r * ldy + o * 8) = *(const uint4*)(A + r * lda + o * 8);
    }
}

template <class E>
__global__ void add_rows_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const svd_bf16* __restrict__ B, int64_t ldb,
                                svd_bf16* __restrict__ Y, int64_t ldy, int64_t rows, int c) {
    const int oct = c >> 3;
    const int64_t total = rows * oct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / oct;
        const int o = (int)(i - r * oct);
        const uint4 a = *(const uint4*)(X + r * ldx + o * 8), b = *(const uint4*)(B + r * ldb + o * 8);
        uint4 w;
        w.x = E::pack(E::lo(a.x) + E::lo(b.x), E::hi(a.x) + E::hi(b.x));
        w.y = E::pack(E::lo(a.y) + E::lo(b.y), E::hi(a.y) + E::hi(b.y));
        w.z = E::pack(E::lo(a.z) + E::lo(b.z), E::hi(a.z) + E::hi(b.z));
        w.w = E::pack(E::lo(a.w) + E::lo(b.w), E::hi(a.w) + E::hi(b.w));
        *(uint4*)(Y + r * ldy + o * 8) = w;
    }
}

// fp32 residual stream: Y(fp32) = X(fp32) + B(16 bit)
template <class E>
__global__ void add_rows_f32_kernel(const float* __restrict__ X, int64_t ldx, const svd_bf16* __restrict__ B, int64_t ldb,
                                    float* __restrict__ Y, int64_t ldy, int64_t rows, int c) {
    const int oct = c >> 3;
    const int64_t total = rows * oct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / oct;
        const int o = (int)(i - r * oct);
        const float* xp = X + r * ldx + o * 8;
        const float4 a0 = *(const float4*)xp, a1 = *(const float4*)(xp + 4);
        const uint4 b = *(const uint4*)(B + r * ldb + o * 8);
        float* yp = Y + r * ldy + o * 8;
        *(float4*)yp = make_float4(a0.x + E::lo(b.x), a0.y + E::hi(b.x), a0.z + E::lo(b.y), a0.w + E::hi(b.y));
        *(float4*)(yp + 4) = make_float4(a1.x + E::lo(b.z), a1.y + E::hi(b.z), a1.z + E::lo(b.w), a1.w + E::hi(b.w));
    }
}

// fp32 rows -> 16-bit rows, 8 channels per thread (the 16-bit operand copy of an fp32 residual-stream tensor: Downsample / Upsample / skip
// convolution inputs, the two halves of the decoder's channel concatenation, the ControlNet features the CAM mergers project)
template <class E>
__global__ void cast_rows_f32_kernel(const float* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y, int64_t ldy, int64_t rows, int c) {
    const int oct = c >> 3;
    const int64_t total = rows * oct;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / oct;
        const int o = (int)(i - r * oct);
        const float* xp = X + r * ldx + o * 8;
        const float4 a0 = *(const float4*)xp, a1 = *(const float4*)(xp + 4);
        uint4 w;
        w.x = E::pack(a0.x, a0.y); w.y = E::pack(a0.z, a0.w); w.z = E::pack(a1.x, a1.y); w.w = E::pack(a1.z, a1.w);
        *(uint4*)(Y + r * ldy + o * 8) = w;
    }
}

// Row permutation of a 4-D array of rows: Y[i_{p0}][i_{p1}][i_{p2}][i_{p3}][:] = X[i0][i1][i2][i3][:], rows of `vec` 16-byte vectors.
// The frame <-> pixel repack of the sequence-parallel all-to-all (parallel.SeqParallel.to_pixels / to_frames) in ONE pass.
__global__ void permute_rows_kernel(const uint4* __restrict__ X, uint4* __restrict__ Y, int n0, int n1, int n2, int n3,
                                    int64_t s0, int64_t s1, int64_t s2, int64_t s3, int vec) {
    // s_k: stride (in rows) of SOURCE dimension k inside the DESTINATION
    const int64_t total = (int64_t)n0 * n1 * n2 * n3 * vec;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / vec;
        const int v = (int)(i - row * vec);
        int64_t q = row;
        const int i3 = (int)(q % n3); q /= n3;
        const int i2 = (int)(q % n2); q /= n2;
        const int i1 = (int)(q % n1); q /= n1;
        const int i0 = (int)q;
        Y[(i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3) * vec + v] = X[i];
    }
}

template <class E>
__global__ void cast_f32_kernel(const float* __restrict__ X, svd_bf16* __restrict__ Y, int64_t n, int apply) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = X[i];
        Y[i] = E::from_f32(apply ? silu_f(x) : x);
    }
}

template <class E, bool F32OUT>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n, int dim, float log_max_period,
                                          void* __restrict__ Yv) {
    const int half = dim >> 1;
    const int64_t total = (int64_t)n * half;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / half), k = (int)(i - (int64_t)r * half);
        const float freq = expf(-log_max_period * (float)k / (float)half);
        const float a = t[r] * freq;
        if constexpr (F32OUT) {          // the precision plan's embedding MLPs take the embedding in fp32 (split-3 operands, csrc/precision.hip)
            float* Y = (float*)Yv;
            Y[(int64_t)r * dim + k] = cosf(a);
            Y[(int64_t)r * dim + half + k] = sinf(a);
            if ((dim & 1) && k == 0) Y[(int64_t)r * dim + dim - 1] = 0.f;
        } else {
            svd_bf16* Y = (svd_bf16*)Yv;
            Y[(int64_t)r * dim + k] = E::from_f32(cosf(a));
            Y[(int64_t)r * dim + half + k] = E::from_f32(sinf(a));
            if ((dim & 1) && k == 0) Y[(int64_t)r * dim + dim - 1] = 0;
        }
    }
}

__global__ void edm_euler_step_kernel(float* __restrict__ x, const float* __restrict__ net, int64_t ldn,
                                      const float* __restrict__ gscale, int T, int C, int pix, float sigma, float sigma_next) {
    const float s2 = sigma * sigma + 1.0f;
    const float c_skip = 1.0f / s2;
    const float c_out = -sigma / sqrtf(s2);
    const float dt = sigma_next - sigma;
    const int64_t total = (int64_t)T * pix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / pix), p = (int)(i - (int64_t)t * pix);
        const float g = gscale[t];
        const float* nu = net + ((int64_t)t * pix + p) * ldn;
        const float* nc = net + ((int64_t)(T + t) * pix + p) * ldn;
        for (int c = 0; c < C; ++c) {
            float* xp = x + ((int64_t)t * C + c) * pix + p;
            const float xv = *xp;
            const float du = nu[c] * c_out + xv * c_skip;
            const float dc = nc[c] * c_out + xv * c_skip;
            const float den = du + g * (dc - du);
            const float d = (xv - den) / sigma;
            *xp = xv + d * dt;
        }
    }
}

__global__ void ae_time_mix3_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ w,
                                    const float* __restrict__ b, float* __restrict__ Y, int frames, int pix, int clamp) {
    __shared__ float sw[27 + 3];
    if (threadIdx.x < 27) sw[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 3) sw[27 + threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    const int64_t total = (int64_t)frames * pix;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i / pix), p = (int)(i - (int64_t)f * pix);
        float o[3] = {sw[27], sw[28], sw[29]};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const int ff = f + kt - 1;
            if (ff < 0 || ff >= frames) continue;
            const float* xr = X + ((int64_t)ff * pix + p) * ldx;
            const float x0 = xr[0], x1 = xr[1], x2 = xr[2];
#pragma unroll
            for (int co = 0; co < 3; ++co)
                o[co] += sw[(co * 3 + 0) * 3 + kt] * x0 + sw[(co * 3 + 1) * 3 + kt] * x1 + sw[(co * 3 + 2) * 3 + kt] * x2;
        }
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            float v = o[co];
            if (clamp) v = fminf(1.f, fmaxf(-1.f, v));
            Y[((int64_t)f * 3 + co) * pix + p] = v;
        }
    }
}

inline unsigned grid_for(int64_t n, int bs = 256) {
    int64_t g = (n + bs - 1) / bs;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// Row A13: the reference's range / quantisation glue, bit for bit.  convert_range([-1,1] -> [0,255]) (utils/result_processor.py:4-14,
// called at diffusion_trainer/streaming_svd.py:353) then IImage(vmin=0, vmax=255) -> torch2np (lib/farancia/libimage/iimage.py:35-36):
//   ((v + 1) / 2) * 255  ->  clip(0, 255)  ->  (255 * x) / 255  ->  NHWC  ->  uint8 by TRUNCATION.
// Every step is a separately rounded fp32 operation in the reference (no FMA contraction): explicit _rn intrinsics.
__global__ __launch_bounds__(256) void frames_to_uint8_kernel(const float* __restrict__ X, uint8_t* __restrict__ Y, int64_t total, int pix) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t f = i / pix, p = i - f * pix;
        const float* x = X + f * 3 * pix + p;
        uint8_t* y = Y + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = x[(int64_t)c * pix];
            v = __fdiv_rn(__fadd_rn(v, 1.0f), 2.0f);
            v = __fmul_rn(v, 255.0f);
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            v = __fdiv_rn(__fmul_rn(255.0f, v), 255.0f);
            y[c] = (uint8_t)v;
        }
    }
}

// exact-erf GELU in place on 16-bit tokens (nn.GELU of the OpenCLIP vision tower's MLP; the UNets use the fused GEGLU epilogue)
template <class E>
__global__ __launch_bounds__(256) void gelu_rows_kernel(svd_bf16* __restrict__ X, int64_t ldx, int64_t rows, int octets) {
    const int64_t total = rows * octets;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / octets; const int o = (int)(i - r * octets);
        uint4* px = (uint4*)(X + r * ldx + o * 8);
        const uint4 u = *px;
        uint4 w;
        w.x = E::pack(gelu_erf_f(E::lo(u.x)), gelu_erf_f(E::hi(u.x))); w.y = E::pack(gelu_erf_f(E::lo(u.y)), gelu_erf_f(E::hi(u.y)));
        w.z = E::pack(gelu_erf_f(E::lo(u.z)), gelu_erf_f(E::hi(u.z))); w.w = E::pack(gelu_erf_f(E::lo(u.w)), gelu_erf_f(E::hi(u.w)));
        *px = w;
    }
}

}  // namespace

extern "C" int svd_nchw_to_tokens(const float* X0, int32_t c0, const float* X1, int32_t c1, const float* scale,
                                  svd_bf16* Y, int32_t cpad, int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream) {
    if (!X0 || !Y || c0 <= 0 || c1 < 0 || (c1 > 0 && !X1) || cpad % 8 || cpad < c0 + c1 || frames <= 0 || pix <= 0) return SVD_EINVAL;
    if ((uintptr_t)Y & 15) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(nchw_to_tokens_kernel<E>, dim3(grid_for((int64_t)frames * pix)), dim3(256), 0,
                                                 (hipStream_t)stream, X0, c0, X1, c1, scale, Y, cpad, frames, pix));
    SVD_CHECK_LAUNCH("nchw_to_tokens");
    return SVD_OK;
}

extern "C" int svd_tokens_to_nchw(const void* X, int32_t x_dtype, int64_t ldx, float* Y, int32_t c, int32_t frames,
                                  int32_t pix, svd_stream_t stream) {
    if (!X || !Y || c <= 0 || frames <= 0 || pix <= 0 || ldx < c) return SVD_EINVAL;
    const int is_f32 = x_dtype == SVD_DTYPE_F32;
    const int dtype = is_f32 ? SVD_DTYPE_BF16 : x_dtype;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(tokens_to_nchw_kernel<E>, dim3(grid_for((int64_t)frames * pix)), dim3(256), 0,
                                                 (hipStream_t)stream, X, is_f32, ldx, Y, c, frames, pix));
    SVD_CHECK_LAUNCH("tokens_to_nchw");
    return SVD_OK;
}

extern "C" int svd_concat_channels(const svd_bf16* A, int64_t lda, int32_t ca, const svd_bf16* B, int64_t ldb, int32_t cb,
                                   svd_bf16* Y, int64_t ldy, int64_t rows, svd_stream_t stream) {
    if (!A || !B || !Y || ca <= 0 || cb <= 0 || ca % 8 || cb % 8 || lda % 8 || ldb % 8 || ldy % 8 || rows <= 0) return SVD_EINVAL;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)Y) & 15) return SVD_EINVAL;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * (ca / 8))), dim3(256), 0, (hipStream_t)stream, A, lda, ca, Y, ldy, rows);
    SVD_CHECK_LAUNCH("concat(A)");
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * (cb / 8))), dim3(256), 0, (hipStream_t)stream, B, ldb, cb, Y + ca, ldy, rows);
    SVD_CHECK_LAUNCH("concat(B)");
    return SVD_OK;
}

extern "C" int svd_add_rows(const void* X, int64_t ldx, const svd_bf16* B, int64_t ldb, void* Y, int64_t ldy,
                            int64_t rows, int32_t channels, int32_t dtype, svd_stream_t stream) {
    if (!X || !B || !Y || channels <= 0 || channels % 8 || ldx % 8 || ldb % 8 || ldy % 8 || rows <= 0) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)B | (uintptr_t)Y) & 15) return SVD_EINVAL;
    if (dtype & SVD_DTYPE_IN_F32) {      // fp32 residual stream: X and Y fp32, B 16 bit
        SVD_DISPATCH_DTYPE(dtype & 0xff, hipLaunchKernelGGL(add_rows_f32_kernel<E>, dim3(grid_for(rows * (channels / 8))), dim3(256), 0,
                                                            (hipStream_t)stream, (const float*)X, ldx, B, ldb, (float*)Y, ldy, rows, channels));
    } else {
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(add_rows_kernel<E>, dim3(grid_for(rows * (channels / 8))), dim3(256), 0,
                                                     (hipStream_t)stream, (const svd_bf16*)X, ldx, B, ldb, (svd_bf16*)Y, ldy, rows, channels));
    }
    SVD_CHECK_LAUNCH("add_rows");
    return SVD_OK;
}

extern "C" int svd_cast_rows_f32(const float* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels, int32_t dtype,
                                 svd_stream_t stream) {
    if (!X || !Y || channels <= 0 || channels % 8 || ldx % 4 || ldy % 8 || rows <= 0) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(cast_rows_f32_kernel<E>, dim3(grid_for(rows * (channels / 8))), dim3(256), 0,
                                                 (hipStream_t)stream, X, ldx, Y, ldy, rows, channels));
    SVD_CHECK_LAUNCH("cast_rows_f32");
    return SVD_OK;
}

extern "C" int svd_permute_rows(const void* X, void* Y, int32_t n0, int32_t n1, int32_t n2, int32_t n3, int32_t p0, int32_t p1, int32_t p2,
                                int32_t p3, int64_t row_bytes, svd_stream_t stream) {
    if (!X || !Y || n0 <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0 || row_bytes <= 0 || row_bytes % 16) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return SVD_EINVAL;
    const int perm[4] = {p0, p1, p2, p3};
    const int64_t n[4] = {n0, n1, n2, n3};
    int seen = 0;
    for (int k = 0; k < 4; ++k) { if (perm[k] < 0 || perm[k] > 3) return SVD_EINVAL; seen |= 1 << perm[k]; }
    if (seen != 15) return SVD_EINVAL;
    // destination dims are n[perm[0..3]]; stride of source dim d inside the destination = product of the destination dims after its position
    int64_t sd[4];
    for (int d = 0; d < 4; ++d) {
        int pos = 0;
        for (int k = 0; k < 4; ++k) if (perm[k] == d) pos = k;
        int64_t st = 1;
        for (int k = pos + 1; k < 4; ++k) st *= n[perm[k]];
        sd[d] = st;
    }
    const int vec = (int)(row_bytes / 16);
    hipLaunchKernelGGL(permute_rows_kernel, dim3(grid_for(n[0] * n[1] * n[2] * n[3] * vec)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)X, (uint4*)Y, n0, n1, n2, n3, sd[0], sd[1], sd[2], sd[3], vec);
    SVD_CHECK_LAUNCH("permute_rows");
    return SVD_OK;
}

extern "C" int svd_cast_f32(const float* X, svd_bf16* Y, int64_t n, int32_t apply_silu, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || n <= 0) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(cast_f32_kernel<E>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, X, Y, n, apply_silu));
    SVD_CHECK_LAUNCH("cast_f32");
    return SVD_OK;
}

extern "C" int svd_timestep_embedding(const float* t, int32_t n, int32_t dim, float max_period, void* Y, int32_t dtype,
                                      svd_stream_t stream) {
    if (!t || !Y || n <= 0 || dim < 2) return SVD_EINVAL;
    if (dtype == SVD_DTYPE_F32) {
        hipLaunchKernelGGL((timestep_embedding_kernel<ElemF16, true>), dim3(grid_for((int64_t)n * (dim / 2))), dim3(256), 0, (hipStream_t)stream, t, n, dim,
                           logf(max_period), Y);
    } else
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((timestep_embedding_kernel<E, false>), dim3(grid_for((int64_t)n * (dim / 2))), dim3(256), 0,
                                                 (hipStream_t)stream, t, n, dim, logf(max_period), Y));
    SVD_CHECK_LAUNCH("timestep_embedding");
    return SVD_OK;
}

extern "C" int svd_edm_euler_step(float* x, const float* net, int64_t ldn, const float* guidance_scale, int32_t T, int32_t C,
                                  int32_t pix, float sigma, float sigma_next, svd_stream_t stream) {
    if (!x || !net || !guidance_scale || T <= 0 || C <= 0 || pix <= 0 || ldn < C || !(sigma > 0.f)) return SVD_EINVAL;
    hipLaunchKernelGGL(edm_euler_step_kernel, dim3(grid_for((int64_t)T * pix)), dim3(256), 0, (hipStream_t)stream, x, net, ldn,
                       guidance_scale, T, C, pix, sigma, sigma_next);
    SVD_CHECK_LAUNCH("edm_euler_step");
    return SVD_OK;
}

extern "C" int svd_ae_time_mix3(const float* X, int64_t ldx, const float* w, const float* b, float* Y, int32_t frames,
                                int32_t pix, int32_t clamp, svd_stream_t stream) {
    if (!X || !w || !b || !Y || frames <= 0 || pix <= 0 || ldx < 3) return SVD_EINVAL;
    hipLaunchKernelGGL(ae_time_mix3_kernel, dim3(grid_for((int64_t)frames * pix)), dim3(256), 0, (hipStream_t)stream, X, ldx, w, b,
                       Y, frames, pix, clamp);
    SVD_CHECK_LAUNCH("ae_time_mix3");
    return SVD_OK;
}

extern "C" int svd_frames_to_uint8(const float* X, uint8_t* Y, int32_t frames, int32_t pix, svd_stream_t stream) {
    if (!X || !Y || frames <= 0 || pix <= 0) return SVD_EINVAL;
    const int64_t total = (int64_t)frames * pix;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(frames_to_uint8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, Y, total, pix);
    SVD_CHECK_LAUNCH("frames_to_uint8");
    return SVD_OK;
}

extern "C" int svd_gelu_rows(svd_bf16* X, int64_t ldx, int64_t rows, int32_t channels, int32_t dtype, svd_stream_t stream) {
    if (!X || rows <= 0 || channels <= 0 || channels % 8 || ldx % 8 || ((uintptr_t)X & 15)) return SVD_EINVAL;
    int64_t blocks = (rows * (channels / 8) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(gelu_rows_kernel<E>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, X, ldx, rows,
                                                 channels / 8));
    SVD_CHECK_LAUNCH("gelu_rows");
    return SVD_OK;
}
