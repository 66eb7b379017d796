// GroupNorm(32) (+SiLU) and LayerNorm on channels-last bf16 activations -- HBM-bound kernels, fp32 statistics.
#include "svd_common.h"
#include <stdlib.h>

namespace {

constexpr int GN_MAXCHUNK = 256;

// 8 consecutive channels of a row as fp32.  IN32 = the fp32 residual stream (dtype | SVD_DTYPE_IN_F32): the norms are the stream's readers,
// their OUTPUT (a GEMM operand) stays 16 bit.
template <class E, bool IN32>
__device__ __forceinline__ void load8(const void* row, int c, float (&v)[8]) {
    if constexpr (IN32) {
        const float4 a = *(const float4*)((const float*)row + c), b = *(const float4*)((const float*)row + c + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4 u = *(const uint4*)((const svd_bf16*)row + c);
        v[0] = E::lo(u.x); v[1] = E::hi(u.x); v[2] = E::lo(u.y); v[3] = E::hi(u.y);
        v[4] = E::lo(u.z); v[5] = E::hi(u.z); v[6] = E::lo(u.w); v[7] = E::hi(u.w);
    }
}
template <bool IN32> __device__ __forceinline__ const void* row_ptr(const void* base, int64_t row, int64_t ld) {
    return IN32 ? (const void*)((const float*)base + row * ld) : (const void*)((const svd_bf16*)base + row * ld);
}

__host__ __device__ inline int gn_nchunk(int frames, int pix) {
    // chunks of ~256 rows, a function of the frame size ONLY: the partial-sum grouping (and so every rounding of the statistics)
    // is then independent of how many frames are batched -- forward(batch 2) == concat(forward(half), forward(half)) bit for
    // bit, which is what makes the CFG-pair split over two GPUs exact (tests/test_gpu_fullsize.py)
    (void)frames;
    int n = (pix > 4096) ? (pix + 255) / 256 : (pix + 63) / 64;      // small frames: smaller chunks keep > 256 workgroups in flight
    if (n > GN_MAXCHUNK) n = GN_MAXCHUNK;
    if (n < 1) n = 1;
    return n;
}

// blockDim = octets * R ; thread -> fixed channel octet (8 channels), rows strided by R.
// partial[f][chunk][g][2] = (sum, sumsq) over the chunk's pixels of group g.
template <class E, bool IN32>
__global__ void gn_stats_partial_kernel(const void* __restrict__ X, int64_t ldx, int pix, int channels, int groups,
                                        int nchunk, float* __restrict__ partial) {
    extern __shared__ float sch[];   // [R][channels][2]: one slot per (row group, channel) -- summed in a FIXED order below
    const int octets = channels >> 3;
    const int R = blockDim.x / octets;
    const int o = threadIdx.x % octets, rr = threadIdx.x / octets;
    const int f = blockIdx.y, chunk = blockIdx.x;
    const int rows_per_chunk = (pix + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > pix) r1 = pix;
    float s[8], ss[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; }
    if (rr < R) {
        const void* base = row_ptr<IN32>(X, (int64_t)f * pix, ldx);
        auto acc8 = [&](const float (&v)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += v[i]; ss[i] += v[i] * v[i]; }
        };
        int r = r0 + rr;
        // (8 rows in flight per thread instead of 4 was measured in round 4: slower, 192 -> 217 us on the level-0 tensor, 305 -> 418 us with fp32 input)
        for (; r + 3 * R < r1; r += 4 * R) {          // 4 independent row loads in flight per thread (HBM-latency bound otherwise)
            float u0[8], u1[8], u2[8], u3[8];
            load8<E, IN32>(row_ptr<IN32>(base, r, ldx), o * 8, u0);
            load8<E, IN32>(row_ptr<IN32>(base, r + R, ldx), o * 8, u1);
            load8<E, IN32>(row_ptr<IN32>(base, r + 2 * R, ldx), o * 8, u2);
            load8<E, IN32>(row_ptr<IN32>(base, r + 3 * R, ldx), o * 8, u3);
            acc8(u0); acc8(u1); acc8(u2); acc8(u3);
        }
        for (; r < r1; r += R) { float u[8]; load8<E, IN32>(row_ptr<IN32>(base, r, ldx), o * 8, u); acc8(u); }
        float* dst = sch + ((int64_t)rr * channels + o * 8) * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dst[2 * i] = s[i]; dst[2 * i + 1] = ss[i]; }
    }
    __syncthreads();
    // deterministic: group g = fixed-order sum over its channels and over the R row groups (no atomics anywhere in the path:
    // bit-identical results from run to run, tests/test_gpu_fullsize.py)
    const int cpg = channels / groups;
    if ((int)threadIdx.x < groups) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < R; ++q)
            for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) {
                a += sch[((int64_t)q * channels + c) * 2]; b += sch[((int64_t)q * channels + c) * 2 + 1];
            }
        float* p = partial + (((int64_t)f * nchunk + chunk) * groups + threadIdx.x) * 2;
        p[0] = a; p[1] = b;
    }
}

// one wave per (stat batch, group): reduce frames_per_stat * nchunk partials -> (mean, rstd)
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nstat, int groups,
                                                         int frames_per_stat, int nchunk, float count, float eps,
                                                         float* __restrict__ stats) {
    const int i = blockIdx.x;
    const int sb = i / groups, g = i - sb * groups;
    const int n = frames_per_stat * nchunk;
    // partial index of entry e (= fr * nchunk + c) : (((sb*fps + fr) * nchunk + c) * groups + g) * 2 = ((sb*n + e) * groups + g) * 2
    const float* base = partial + ((int64_t)sb * n * groups + g) * 2;
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < n; e += 64) {
        const float2 v = *(const float2*)(base + (int64_t)e * groups * 2);
        a += v.x; b += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (threadIdx.x == 0) {
        const double mean = a / count;
        double var = b / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[2 * i + 0] = (float)mean;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// sequence-parallel form of the 5-D statistics: the same fixed-order reduction of the partials, but the (sum, sum of squares) pair is
// written out in double so that the ranks of a sequence-parallel group can add their shares (RCCL all-reduce) before the statistics
// are formed by gn_stats_from_sums_kernel with the GLOBAL element count.
__global__ __launch_bounds__(64) void gn_sums_kernel(const float* __restrict__ partial, int groups, int frames_per_stat, int nchunk,
                                                     double* __restrict__ sums) {
    const int i = blockIdx.x;
    const int sb = i / groups, g = i - sb * groups;
    const int n = frames_per_stat * nchunk;
    const float* base = partial + ((int64_t)sb * n * groups + g) * 2;
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < n; e += 64) {
        const float2 v = *(const float2*)(base + (int64_t)e * groups * 2);
        a += v.x; b += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (threadIdx.x == 0) { sums[2 * i + 0] = a; sums[2 * i + 1] = b; }
}

__global__ void gn_stats_from_sums_kernel(const double* __restrict__ sums, int n, double count, float eps, float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double mean = sums[2 * i] / count;
    double var = sums[2 * i + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * i + 0] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// FIN = the statistics are FINALIZED HERE (round 4): every workgroup reduces the frames_per_stat * nchunk partial sums of its (stat batch,
// group) pairs itself -- at most 64 entries per group, i.e. <= 16 KB of L2-resident partials against the 100+ KB of activations the workgroup
// moves -- instead of reading the output of a separate gn_finalize launch (15.5 k launches of 5 us per three AR chunks, round-3 kernel trace;
// the time_stack / CAM norms, which pool 25 frames x 36 chunks, keep the finalize kernel).  Same arithmetic (double sums, fixed order).
template <class E, bool IN32, bool FIN>
__global__ void gn_apply_kernel(const void* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y, int64_t ldy, int pix,
                                int channels, int groups, int frames_per_stat, int nchunk, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int silu, float count, float eps) {
    const int octets = channels >> 3;
    const int R = blockDim.x / octets;
    const int o = threadIdx.x % octets, rr = threadIdx.x / octets;
    const int f = blockIdx.y, chunk = blockIdx.x;
    const int rows_per_chunk = (pix + nchunk - 1) / nchunk;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > pix) r1 = pix;
    const int cpg = channels / groups;
    const int sb = f / frames_per_stat;
    __shared__ float sst[64];
    if constexpr (FIN) {
        // `stats` is the PARTIAL buffer here: entry e = fr * nchunk + c of (sb, g) at ((sb * n + e) * groups + g) * 2.  Two threads per group
        // (even / odd entries), combined in a fixed order.
        const int t = threadIdx.x;
        if (t < 2 * groups) {
            const int g = t >> 1, half = t & 1;
            const int n = frames_per_stat * nchunk;
            const float* base = stats + ((int64_t)sb * n * groups + g) * 2;
            double a = 0.0, b = 0.0;
            for (int e = half; e < n; e += 2) {
                const float2 v = *(const float2*)(base + (int64_t)e * groups * 2);
                a += v.x; b += v.y;
            }
            a += __shfl_xor(a, 1, 64); b += __shfl_xor(b, 1, 64);
            if (!half) {
                const double mean = a / (double)count;
                double var = b / (double)count - mean * mean;
                if (var < 0.0) var = 0.0;
                sst[2 * g] = (float)mean;
                sst[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
            }
        }
        __syncthreads();
    }
    if (rr >= R) return;
    float ca[8], cb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = o * 8 + i;
        const int g = c / cpg;
        float mean, rstd;
        if constexpr (FIN) { mean = sst[2 * g]; rstd = sst[2 * g + 1]; }
        else { mean = stats[((int64_t)sb * groups + g) * 2]; rstd = stats[((int64_t)sb * groups + g) * 2 + 1]; }
        const float ga = gamma[c] * rstd;
        ca[i] = ga; cb[i] = beta[c] - mean * ga;
    }
    const void* xb = row_ptr<IN32>(X, (int64_t)f * pix, ldx);
    svd_bf16* yb = Y + ((int64_t)f * pix) * ldy + o * 8;
    auto one = [&](float (&v)[8], int r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = v[i] * ca[i] + cb[i];
            if (silu) v[i] = silu_f(v[i]);
        }
        uint4 w;
        w.x = E::pack(v[0], v[1]); w.y = E::pack(v[2], v[3]);
        w.z = E::pack(v[4], v[5]); w.w = E::pack(v[6], v[7]);
        *(uint4*)(yb + (int64_t)r * ldy) = w;
    };
    int r = r0 + rr;
    for (; r + 3 * R < r1; r += 4 * R) {              // 4 independent loads in flight per thread
        float u0[8], u1[8], u2[8], u3[8];
        load8<E, IN32>(row_ptr<IN32>(xb, r, ldx), o * 8, u0);
        load8<E, IN32>(row_ptr<IN32>(xb, r + R, ldx), o * 8, u1);
        load8<E, IN32>(row_ptr<IN32>(xb, r + 2 * R, ldx), o * 8, u2);
        load8<E, IN32>(row_ptr<IN32>(xb, r + 3 * R, ldx), o * 8, u3);
        one(u0, r); one(u1, r + R); one(u2, r + 2 * R); one(u3, r + 3 * R);
    }
    for (; r < r1; r += R) { float u[8]; load8<E, IN32>(row_ptr<IN32>(xb, r, ldx), o * 8, u); one(u, r); }
}

// LayerNorm: one wave per token row, NR rows per wave in flight (the kernel is latency-bound with one 16-byte load per lane
// per row: 3.6 TB/s measured); up to MAXV 16-byte vectors per lane (C <= 64*8*MAXV).
template <int MAXV, class E, bool IN32>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y,
                                                        int64_t ldy, int64_t rows, int channels, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        const float* __restrict__ addvec, int addvec_ld, int rows_per_vec,
                                                        void* __restrict__ Xsum, int64_t ldxsum, int silu) {
    constexpr int NR = (MAXV <= 2) ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int octets = channels >> 3;
    const float invc = 1.f / (float)channels;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row0 = wave_id * NR; row0 < rows; row0 += nwaves * NR) {
        float v[NR][MAXV][8];
        float sum[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            sum[j] = 0.f;
            int64_t row = row0 + j; if (row > rows - 1) row = rows - 1;            // tail: recompute the last row, store masked below
            const void* xr = row_ptr<IN32>(X, row, ldx);
            const float* av = addvec ? addvec + (row / rows_per_vec) * addvec_ld : nullptr;
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
                const int o = lane + 64 * k;
                if (o < octets) {
                    load8<E, IN32>(xr, o * 8, v[j][k]);
                    if (av) {
                        const float4 a0 = *(const float4*)(av + o * 8), a1 = *(const float4*)(av + o * 8 + 4);
                        v[j][k][0] += a0.x; v[j][k][1] += a0.y; v[j][k][2] += a0.z; v[j][k][3] += a0.w;
                        v[j][k][4] += a1.x; v[j][k][5] += a1.y; v[j][k][6] += a1.z; v[j][k][7] += a1.w;
                        if (Xsum && row0 + j < rows) {
                            if constexpr (IN32) {          // the sum continues the fp32 residual stream
                                float* xs = (float*)Xsum + row * ldxsum + o * 8;
                                *(float4*)xs = make_float4(v[j][k][0], v[j][k][1], v[j][k][2], v[j][k][3]);
                                *(float4*)(xs + 4) = make_float4(v[j][k][4], v[j][k][5], v[j][k][6], v[j][k][7]);
                            } else {
                                uint4 w;
                                w.x = E::pack(v[j][k][0], v[j][k][1]); w.y = E::pack(v[j][k][2], v[j][k][3]);
                                w.z = E::pack(v[j][k][4], v[j][k][5]); w.w = E::pack(v[j][k][6], v[j][k][7]);
                                *(uint4*)((svd_bf16*)Xsum + row * ldxsum + o * 8) = w;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) sum[j] += v[j][k][i];
                }
            }
        }
        float mean[NR], rstd[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) mean[j] = wave_sum(sum[j]) * invc;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < MAXV; ++k) {
                const int o = lane + 64 * k;
                if (o < octets) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const float d = v[j][k][i] - mean[j]; sq += d * d; }
                }
            }
            rstd[j] = rsqrtf(wave_sum(sq) * invc + eps);
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int o = lane + 64 * k;
            if (o < octets) {
                const float4 g0 = *(const float4*)(gamma + o * 8), g1 = *(const float4*)(gamma + o * 8 + 4);
                const float4 b0 = *(const float4*)(beta + o * 8), b1 = *(const float4*)(beta + o * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    if (row0 + j >= rows) continue;
                    float y[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        y[i] = (v[j][k][i] - mean[j]) * rstd[j] * gg[i] + bb[i];
                        if (silu) y[i] = silu_f(y[i]);
                    }
                    uint4 w;
                    w.x = E::pack(y[0], y[1]); w.y = E::pack(y[2], y[3]);
                    w.z = E::pack(y[4], y[5]); w.w = E::pack(y[6], y[7]);
                    *(uint4*)(Y + (row0 + j) * ldy + o * 8) = w;
                }
            }
        }
    }
}

// LayerNorm, PACKED form (round 4): LPR lanes per token row, 64 / LPR rows per wave, up to 5 16-byte vectors per lane (octet = sub + LPR * k,
// so the LPR lanes of a row read LPR x 16 contiguous bytes per instruction).  The one-row-per-wave kernel above leaves 24 of 64 lanes idle at
// C = 320 (40 octets) and has ONE 16-byte load per lane in flight per row: 3.1-3.4 TB/s at C = 320 / 640 against 5.0 TB/s for its own fp32-input
// form (profiles/r03_norm_bandwidth.txt).  Here every lane carries data at C = 320 / 640 / 1280 (LPR = 8 / 16 / 32, 5 vectors each) and a wave
// has 5 x 1 KiB of loads in flight.  Same arithmetic as above (two passes over the registers, fp32); the reductions run inside the LPR lanes.
template <int LPR> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int LPR, class E, bool IN32>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(const void* __restrict__ X, int64_t ldx, svd_bf16* __restrict__ Y,
                                                               int64_t ldy, int64_t rows, int channels, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               const float* __restrict__ addvec, int addvec_ld, int rows_per_vec,
                                                               void* __restrict__ Xsum, int64_t ldxsum, int silu) {
    constexpr int MAXV = 5;
    constexpr int RW = 64 / LPR;                       // rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR, rsel = lane / LPR;
    const int octets = channels >> 3;
    const float invc = 1.f / (float)channels;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row0 = wave_id * RW; row0 < rows; row0 += nwaves * RW) {
        const bool live = row0 + rsel < rows;
        const int64_t row = live ? row0 + rsel : rows - 1;                // tail: recompute the last row, stores masked
        const void* xr = row_ptr<IN32>(X, row, ldx);
        const float* av = addvec ? addvec + (row / rows_per_vec) * addvec_ld : nullptr;
        float v[MAXV][8];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int o = sub + LPR * k;
            if (o < octets) load8<E, IN32>(xr, o * 8, v[k]);
        }
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int o = sub + LPR * k;
            if (o < octets) {
                if (av) {
                    const float4 a0 = *(const float4*)(av + o * 8), a1 = *(const float4*)(av + o * 8 + 4);
                    v[k][0] += a0.x; v[k][1] += a0.y; v[k][2] += a0.z; v[k][3] += a0.w;
                    v[k][4] += a1.x; v[k][5] += a1.y; v[k][6] += a1.z; v[k][7] += a1.w;
                    if (Xsum && live) {
                        if constexpr (IN32) {          // the sum continues the fp32 residual stream
                            float* xs = (float*)Xsum + row * ldxsum + o * 8;
                            *(float4*)xs = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
                            *(float4*)(xs + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
                        } else {
                            uint4 w;
                            w.x = E::pack(v[k][0], v[k][1]); w.y = E::pack(v[k][2], v[k][3]);
                            w.z = E::pack(v[k][4], v[k][5]); w.w = E::pack(v[k][6], v[k][7]);
                            *(uint4*)((svd_bf16*)Xsum + row * ldxsum + o * 8) = w;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) sum += v[k][i];
            }
        }
        const float mean = group_sum<LPR>(sum) * invc;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int o = sub + LPR * k;
            if (o < octets) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float d = v[k][i] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(group_sum<LPR>(sq) * invc + eps);
#pragma unroll
        for (int k = 0; k < MAXV; ++k) {
            const int o = sub + LPR * k;
            if (o < octets && live) {
                const float4 g0 = *(const float4*)(gamma + o * 8), g1 = *(const float4*)(gamma + o * 8 + 4);
                const float4 b0 = *(const float4*)(beta + o * 8), b1 = *(const float4*)(beta + o * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y[i] = (v[k][i] - mean) * rstd * gg[i] + bb[i];
                    if (silu) y[i] = silu_f(y[i]);
                }
                uint4 w;
                w.x = E::pack(y[0], y[1]); w.y = E::pack(y[2], y[3]);
                w.z = E::pack(y[4], y[5]); w.w = E::pack(y[6], y[7]);
                *(uint4*)(Y + row * ldy + o * 8) = w;
            }
        }
    }
}

inline int gn_block(int channels) {
    const int octets = channels >> 3;
    int R = 256 / octets; if (R < 1) R = 1;
    return octets * R;
}

}  // namespace

extern "C" int64_t svd_groupnorm_partial_elems(int32_t frames, int32_t channels) {
    (void)channels;
    return (int64_t)frames * GN_MAXCHUNK * 64;
}

#define SVD_DISPATCH_IN(dtype, ...)                                                              \
    do {                                                                                         \
        const int base_dt__ = (dtype) & 0xff;                                                    \
        if ((dtype) & SVD_DTYPE_IN_F32) { constexpr bool IN32 = true; SVD_DISPATCH_DTYPE(base_dt__, __VA_ARGS__); } \
        else { constexpr bool IN32 = false; SVD_DISPATCH_DTYPE(base_dt__, __VA_ARGS__); }        \
    } while (0)

extern "C" int svd_groupnorm_stats(const void* X, int64_t ldx, int32_t frames, int32_t pix, int32_t channels,
                                   int32_t groups, int32_t frames_per_stat, float eps, float* partial, float* stats,
                                   int32_t dtype, svd_stream_t stream) {
    if (!X || !partial || !stats || frames <= 0 || pix <= 0 || channels <= 0) return SVD_EINVAL;
    if (groups <= 0 || groups > 32 || channels % groups || channels % 8 || ldx % 8 || channels > 8192) return SVD_EINVAL;
    if (frames_per_stat <= 0 || frames % frames_per_stat || frames > 65535) return SVD_EINVAL;
    if ((uintptr_t)X & 15) return SVD_EINVAL;
    const int nchunk = gn_nchunk(frames, pix);
    const int bs = gn_block(channels);
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((gn_stats_partial_kernel<E, IN32>), dim3(nchunk, frames), dim3(bs), (size_t)(bs / (channels >> 3)) * 2 * channels * sizeof(float),
                                              (hipStream_t)stream, X, ldx, pix, channels, groups, nchunk, partial));
    SVD_CHECK_LAUNCH("gn_stats_partial");
    const int nstat = frames / frames_per_stat;
    const float count = (float)frames_per_stat * (float)pix * (float)(channels / groups);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nstat * groups), dim3(64), 0, (hipStream_t)stream, partial,
                       nstat, groups, frames_per_stat, nchunk, count, eps, stats);
    SVD_CHECK_LAUNCH("gn_finalize");
    return SVD_OK;
}

extern "C" int svd_groupnorm_sums(const void* X, int64_t ldx, int32_t frames, int32_t pix, int32_t channels, int32_t groups,
                                  int32_t frames_per_stat, float* partial, double* sums, int32_t dtype, svd_stream_t stream) {
    if (!X || !partial || !sums || frames <= 0 || pix <= 0 || channels <= 0) return SVD_EINVAL;
    if (groups <= 0 || groups > 32 || channels % groups || channels % 8 || ldx % 8 || channels > 8192) return SVD_EINVAL;
    if (frames_per_stat <= 0 || frames % frames_per_stat || frames > 65535) return SVD_EINVAL;
    if (((uintptr_t)X & 15) || ((uintptr_t)sums & 7)) return SVD_EINVAL;
    const int nchunk = gn_nchunk(frames, pix);
    const int bs = gn_block(channels);
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((gn_stats_partial_kernel<E, IN32>), dim3(nchunk, frames), dim3(bs), (size_t)(bs / (channels >> 3)) * 2 * channels * sizeof(float),
                                              (hipStream_t)stream, X, ldx, pix, channels, groups, nchunk, partial));
    SVD_CHECK_LAUNCH("gn_stats_partial");
    hipLaunchKernelGGL(gn_sums_kernel, dim3((frames / frames_per_stat) * groups), dim3(64), 0, (hipStream_t)stream, partial, groups, frames_per_stat,
                       nchunk, sums);
    SVD_CHECK_LAUNCH("gn_sums");
    return SVD_OK;
}

extern "C" int svd_groupnorm_stats_from_sums(const double* sums, int32_t nstat, int32_t groups, double count, float eps, float* stats,
                                             svd_stream_t stream) {
    if (!sums || !stats || nstat <= 0 || groups <= 0 || groups > 32 || !(count > 0.0)) return SVD_EINVAL;
    const int n = nstat * groups;
    hipLaunchKernelGGL(gn_stats_from_sums_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, n, count, eps, stats);
    SVD_CHECK_LAUNCH("gn_stats_from_sums");
    return SVD_OK;
}

extern "C" int svd_groupnorm_apply(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t pix,
                                   int32_t channels, int32_t groups, int32_t frames_per_stat, const float* stats,
                                   const float* gamma, const float* beta, int32_t silu, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || !stats || !gamma || !beta || frames <= 0 || pix <= 0) return SVD_EINVAL;
    if (groups <= 0 || groups > 32 || channels % groups || channels % 8 || ldx % 8 || ldy % 8 || channels > 8192) return SVD_EINVAL;
    if (frames_per_stat <= 0 || frames % frames_per_stat || frames > 65535) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return SVD_EINVAL;
    const int nchunk = gn_nchunk(frames, pix);
    const int bs = gn_block(channels);
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((gn_apply_kernel<E, IN32, false>), dim3(nchunk, frames), dim3(bs), 0, (hipStream_t)stream, X, ldx, Y, ldy,
                                              pix, channels, groups, frames_per_stat, nchunk, stats, gamma, beta, silu, 0.f, 0.f));
    SVD_CHECK_LAUNCH("gn_apply");
    return SVD_OK;
}

// GroupNorm (+SiLU) in one call (round 4): statistics pass + apply pass; the apply pass finalizes the statistics itself when a (stat batch,
// group) pair has <= 64 partial sums (every per-frame norm of the UNet / ControlNet), else gn_finalize runs in between as before.
// Replaces torch.nn.GroupNorm / GroupNorm32 (util.py:274-276) like the two-call form; `stats` is only written on the finalize path.
extern "C" int svd_groupnorm(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t pix, int32_t channels,
                             int32_t groups, int32_t frames_per_stat, float eps, float* partial, float* stats, const float* gamma,
                             const float* beta, int32_t silu, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || !partial || !stats || !gamma || !beta || frames <= 0 || pix <= 0 || channels <= 0) return SVD_EINVAL;
    if (groups <= 0 || groups > 32 || channels % groups || channels % 8 || ldx % 8 || ldy % 8 || channels > 8192) return SVD_EINVAL;
    if (frames_per_stat <= 0 || frames % frames_per_stat || frames > 65535) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Y) & 15) return SVD_EINVAL;
    const int nchunk = gn_nchunk(frames, pix);
    const int bs = gn_block(channels);
    static const bool fuse = []{ const char* e = getenv("SVD_GN_FUSED_FINALIZE"); return !(e && e[0] == '0'); }();
    if (!fuse || frames_per_stat * nchunk > 64 || bs < 2 * groups) {
        int rc = svd_groupnorm_stats(X, ldx, frames, pix, channels, groups, frames_per_stat, eps, partial, stats, dtype, stream);
        if (rc != SVD_OK) return rc;
        return svd_groupnorm_apply(X, ldx, Y, ldy, frames, pix, channels, groups, frames_per_stat, stats, gamma, beta, silu, dtype, stream);
    }
    const float count = (float)frames_per_stat * (float)pix * (float)(channels / groups);
    // (Running the two passes in frame batches sized for the 256-MiB Infinity Cache, last-written frames first, was measured and is slower at every
    // batch size: profiles/r04_groupnorm_frame_batches.txt.)
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((gn_stats_partial_kernel<E, IN32>), dim3(nchunk, frames), dim3(bs), (size_t)(bs / (channels >> 3)) * 2 * channels * sizeof(float),
                                              (hipStream_t)stream, X, ldx, pix, channels, groups, nchunk, partial));
    SVD_CHECK_LAUNCH("gn_stats_partial");
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((gn_apply_kernel<E, IN32, true>), dim3(nchunk, frames), dim3(bs), 0, (hipStream_t)stream, X, ldx, Y, ldy,
                                              pix, channels, groups, frames_per_stat, nchunk, partial, gamma, beta, silu, count, eps));
    SVD_CHECK_LAUNCH("gn_apply_fin");
    return SVD_OK;
}

extern "C" int svd_layernorm(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels,
                             const float* gamma, const float* beta, float eps, const float* addvec, int32_t addvec_ld,
                             int32_t rows_per_vec, void* Xsum, int64_t ldxsum, int32_t silu, int32_t dtype, svd_stream_t stream) {
    if (!X || !Y || !gamma || !beta || rows <= 0 || channels <= 0 || channels % 8 || ldx % 8 || ldy % 8) return SVD_EINVAL;
    if (channels > 64 * 8 * 4) return SVD_EINVAL;
    if (addvec && (rows_per_vec <= 0 || addvec_ld % 4)) return SVD_EINVAL;
    if (Xsum && (!addvec || ldxsum % 8)) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)addvec | (uintptr_t)Xsum) & 15) return SVD_EINVAL;
    const int octets = channels / 8;
    // packed form (LPR lanes per row, <= 5 vectors per lane) for C <= 1280; SVD_LN_PACKED=0: the one-row-per-wave kernel (A/B reference)
    static const bool packed = []{ const char* e = getenv("SVD_LN_PACKED"); return !(e && e[0] == '0'); }();
    // (measured, gpurun r4a: C = 320 181 -> 150 us, C = 640 96 -> 81 us for the level's 295 MB; at C = 1280 the one-row-per-wave kernel already has 3
    // vectors per lane in flight and is as fast or faster (33.2 vs 33.9 us; fp32 input 38.5 vs 43.0) -> packed form for C <= 640 only)
    if (packed && octets <= 80) {
        const int lpr = octets <= 20 ? 4 : octets <= 40 ? 8 : 16;
        const int rows_per_block = 4 * (64 / lpr);
        int64_t nb = (rows + rows_per_block - 1) / rows_per_block;          // one pass per workgroup (a capped grid left 1.76 passes per workgroup at
        if (nb > 65535 * 8) nb = 65535 * 8;                                  // level 0: half the workgroups ran twice as long as the others)
#define LNP_LAUNCH(L)                                                                                                                \
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((layernorm_packed_kernel<L, E, IN32>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, \
                                              X, ldx, Y, ldy, rows, channels, gamma, beta, eps, addvec, addvec_ld, rows_per_vec,            \
                                              Xsum, ldxsum, silu))
        if (lpr == 4) LNP_LAUNCH(4);
        else if (lpr == 8) LNP_LAUNCH(8);
        else LNP_LAUNCH(16);
#undef LNP_LAUNCH
        SVD_CHECK_LAUNCH("layernorm_packed");
        return SVD_OK;
    }
    int64_t blocks = (rows + 7) / 8;                      // 4 waves x (up to) 2 rows per workgroup pass
    if (blocks > 256 * 32) blocks = 256 * 32;
#define LN_LAUNCH(MV)                                                                                                        \
    SVD_DISPATCH_IN(dtype, hipLaunchKernelGGL((layernorm_kernel<MV, E, IN32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, \
                                              X, ldx, Y, ldy, rows, channels, gamma, beta, eps, addvec, addvec_ld, rows_per_vec,    \
                                              Xsum, ldxsum, silu))
    if (octets <= 64) LN_LAUNCH(1);
    else if (octets <= 128) LN_LAUNCH(2);
    else if (octets <= 192) LN_LAUNCH(3);
    else LN_LAUNCH(4);
#undef LN_LAUNCH
    SVD_CHECK_LAUNCH("layernorm");
    return SVD_OK;
}
