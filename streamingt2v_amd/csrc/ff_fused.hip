// Fused GEGLU feed-forward for the 320-channel level (round 5):   Y = R + b2 + W2 . ( (W1v X + b1v) * gelu_erf(W1g X + b1g) )
//
// Replaces the TWO launches of FeedForward (code/models/svd/sgm/modules/attention.py:94-120: GEGLU.proj + gate, then net[2]; diffusers
// FeedForward(activation_fn="geglu") in the enhancer, code/i2v_enhance/attention.py:414-534) for dim = 320, inner = 1280 -- the level-0 blocks
// of the VideoUNet / ControlNet / I2VGenXLUNet, where the [M, 1280] hidden tensor is the largest byte-mover of a forward (1.18 GB written by
// the projection and read straight back by the down-projection at M = 460 800).  Here the hidden tile never leaves the CU:
//
//   * one WAVE owns 32 token rows for the whole computation; 4 waves = one 128-row workgroup tile, one wave per SIMD (the accumulators of
//     the full 320-channel output row block are 160 registers per lane);
//   * the 32 x 320 input rows of the wave stay in 80 VGPRs as the B operands of   S^T[hidden, row] = W1 . X^T   (A = W1 fragments from LDS);
//     a lane therefore holds, for ITS row, 8 value and the 8 matching gate pre-activations of every 16-hidden MFMA tile (the packed weight
//     rows of a tile are 16 value rows followed by the 16 gate rows of the same hidden units) -> GEGLU is lane-local;
//   * the packed 16-bit GEGLU result of a tile IS the B-operand fragment of   O^T[channel, row] += W2 . H^T   (the hidden order inside a
//     16-unit k-step is whatever order the accumulator registers have; W2 is packed on the host in the same order) -- the attention
//     kernels' "P never leaves registers" applied to the feed-forward;
//   * the weights (2.4 MB: L2-resident) stream through LDS in chunks of 32 hidden units (41 KiB of W1 fragments + biases, 20 KiB of W2
//     fragments) as LDS-DMA copies of a host-packed fragment-order image: a fragment is one conflict-free ds_read_b128;
//   * software pipeline inside the wave: the 40 MFMAs of S^T for chunk c+1 are issued with the GELU arithmetic of chunk c (two accumulator
//     sets swap roles), then the 20 MFMAs of O^T for chunk c.
// Work per 128-row tile and wave: 40 chunks x 60 MFMA 32x32x16 = 76.8 k matrix-pipe cycles; HBM traffic: X read once, R read once, Y written once.
#include "svd_common.h"

namespace {

constexpr int FF_C = 320;                        // channels of the level: K of GEMM 1, N of GEMM 2
constexpr int FF_NS = FF_C / 16;                 // 20 k-steps of GEMM 1
constexpr int FF_NO = FF_C / 32;                 // 10 output tiles of GEMM 2
constexpr int FF_W1_PIECES = 2 * FF_NS + 1;      // 40 fragments of 1 KiB + 1 KiB holding the chunk's 64 biases
constexpr int FF_W2_PIECES = 2 * FF_NO;          // 20 fragments
constexpr int FF_W1_BYTES = FF_W1_PIECES * 1024, FF_W2_BYTES = FF_W2_PIECES * 1024;
constexpr int FF_BLOB = FF_W1_BYTES + FF_W2_BYTES;             // one chunk (32 hidden units) of the packed image: 61 KiB
constexpr int FF_LDS_W2 = 2 * FF_W1_BYTES;                     // LDS: W1 ring (2 slots) | W2 ring (2 slots) | b2
constexpr int FF_LDS_B2 = FF_LDS_W2 + 2 * FF_W2_BYTES;
constexpr int FF_LDS_TOTAL = FF_LDS_B2 + FF_C * 4;             // 126 208 B

// RES: 0 = no residual, 1 = 16-bit residual rows, 2 = fp32 residual rows (the fp32 residual stream); OUT32: fp32 output rows;
// BLEND: Y = alpha * S + (1 - alpha) * (...) with S of the residual's type (the temporal block's AlphaBlender, video_attention.py:318-322)
// PV (probe builds only, -DSVD_FF_PROBES; results are WRONG for PV != 0): 1 = no LDS-DMA in the steps (stale weights), 2 = no GELU arithmetic,
// 3 = neither, 4 = no MFMA of S^T (phase A: reads + GELU + DMA only)
template <class E, int RES, bool OUT32, bool BLEND = false, int PV = 0>
__global__ __launch_bounds__(256, 1) void ff_geglu_fused_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const char* __restrict__ Wp, int nch,
                                                                const float* __restrict__ b2, const void* __restrict__ R, int64_t ldr,
                                                                const void* __restrict__ S, int64_t lds, float alpha,
                                                                void* __restrict__ Y, int64_t ldy, int M, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t voff = (uint32_t)lane * 16u;

    // ---- LDS-DMA of one chunk's W1 / W2 image: piece p (1 KiB) by wave p % 4; every wave issues the same number of instructions (11 + 5) so that
    //      the vmcnt bookkeeping is wave-uniform (waves 1..3 re-copy piece 40 once: same bytes to the same place)
    // (asm volatile WITHOUT a "memory" clobber: the copies target a ring slot nobody reads in this step -- its readers are fenced off by the
    //  workgroup barriers either side -- so the compiler may move this step's fragment reads freely around them; volatile keeps them in order
    //  with the explicit waits and barriers)
    auto glds = [&](const char* src, uint32_t dst) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(dst) : "m0");
    };
    auto dma_w1 = [&](int ch, int slot) {
        const char* src = Wp + (int64_t)ch * FF_BLOB;
        const uint32_t dst = sbase + slot * FF_W1_BYTES;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            int p = wave + 4 * i;
            p = p > FF_W1_PIECES - 1 ? FF_W1_PIECES - 1 : p;
            glds(src + p * 1024, dst + p * 1024);
        }
    };
    auto dma_w2 = [&](int ch, int slot) {
        const char* src = Wp + (int64_t)ch * FF_BLOB + FF_W1_BYTES;
        const uint32_t dst = sbase + FF_LDS_W2 + slot * FF_W2_BYTES;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int p = wave + 4 * i;
            glds(src + p * 1024, dst + p * 1024);
        }
    };

    // ---- X fragments of the wave's 32 rows: lane (row l31, half hi) holds channels 16 s + 8 hi .. + 7 of k-step s (the B operand of GEMM 1)
    uint4 xf[FF_NS];
    auto load_x = [&](int tile) {
        int row = tile * 128 + wave * 32 + l31;
        row = row < M ? row : M - 1;
        const svd_bf16* xp = X + (int64_t)row * ldx + 8 * hi;
#pragma unroll
        for (int s = 0; s < FF_NS; ++s) xf[s] = *(const uint4*)(xp + 16 * s);
    };
    // ---- S^T of one chunk: accumulators start from the biases (rows 8 j + 4 hi + q of each 32-row tile: registers 4 j + q)
    auto bias_init = [&](const char* w, f32x16_t (&s_acc)[2]) __attribute__((always_inline)) {
        const char* bp = w + 2 * FF_NS * 1024 + hi * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 b = *(const float4*)(bp + t * 128 + j * 32);
                s_acc[t][4 * j + 0] = b.x; s_acc[t][4 * j + 1] = b.y; s_acc[t][4 * j + 2] = b.z; s_acc[t][4 * j + 3] = b.w;
            }
    };
    // the tile's first chunk: nothing to overlap with (once per 40 chunks); fragments three MFMAs ahead through a ring of four registers
    auto gemm1_plain = [&](int slot, f32x16_t (&s_acc)[2]) __attribute__((always_inline)) {
        const char* w = smem + slot * FF_W1_BYTES;
        bias_init(w, s_acc);
        const char* wl = w + lane * 16;
        uint4 fr[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) fr[i] = *(const uint4*)(wl + i * 1024);
#pragma unroll
        for (int i = 0; i < 2 * FF_NS; ++i) {
            if (i + 3 < 2 * FF_NS) fr[(i + 3) & 3] = *(const uint4*)(wl + (i + 3) * 1024);
            s_acc[i & 1] = E::mfma(fr[i & 3], xf[i >> 1], s_acc[i & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16_t o_acc[FF_NO];

    // ---- prologue: b2 -> LDS, the first two W1 chunks and the first W2 chunk, the first tile's rows
    for (int i = tid; i < FF_C; i += 256) ((float*)(smem + FF_LDS_B2))[i] = b2[i];
    int tile = blockIdx.x;
    if (tile < ntiles) {
        dma_w1(0, 0);
        dma_w1(1 % nch, 1);
        dma_w2(0, 0);
        load_x(tile);
    }
    svd_wait_dma();
    __syncthreads();

    f32x16_t s_a[2], s_b[2];
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int o = 0; o < FF_NO; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[o][i] = 0.f;
        gemm1_plain(0, s_a);                 // chunk 0 of this tile (W1 slot 0: nch is even, so every tile starts on slot parity 0)
        __syncthreads();                     // every wave is done with W1 slot 0 before iteration 0 overwrites it with chunk 2
        // One chunk, hand-placed (one wave per SIMD: nothing but this wave's own instruction order hides latency).  `sc` holds S^T of chunk c
        // (computed one step earlier), `sn` receives S^T of chunk c + 1 from W1 slot PAR ^ 1.
        //   phase A, 40 slots: MFMA i of S^T(c + 1) | the fragment read for MFMA i + 3 | one fifth of a GELU pair of chunk c (8 pairs x 5 stages)
        //                      | the step's 16 LDS-DMA pieces at slots 1, 3 (mod 5)
        //   phase B, 20 slots: MFMA j of O^T(c) | the fragment read for MFMA j + 3 | (LOADX) one row load of the next tile
        // A sched_barrier after every slot pins the order; LAST / LOADX / PAR are compile-time (no branches inside the phases).
        auto step = [&]<int PAR, bool LAST, bool LOADX>(int c, f32x16_t (&sc)[2], f32x16_t (&sn)[2]) __attribute__((always_inline)) {
            int c2 = c + 2; c2 = c2 >= nch ? c2 - nch : c2;
            int c1 = c + 1; c1 = c1 >= nch ? 0 : c1;
            // chunk c + 2 (of this tile, or chunks 0 / 1 of the next) goes into the W1 slot chunk c was read from one step ago; W2 of chunk c + 1 likewise
            const char* src1 = Wp + (int64_t)c2 * FF_BLOB;
            const char* src2 = Wp + (int64_t)c1 * FF_BLOB + FF_W1_BYTES;
            const uint32_t dst1 = sbase + PAR * FF_W1_BYTES, dst2 = sbase + FF_LDS_W2 + (PAR ^ 1) * FF_W2_BYTES;
            auto dma = [&](int k) __attribute__((always_inline)) {          // piece k of this wave's 16 (11 of W1: wave + 4 k clamped to the last piece; 5 of W2)
                if (k < 11) {
                    int p = wave + 4 * k;
                    p = p > FF_W1_PIECES - 1 ? FF_W1_PIECES - 1 : p;
                    glds(src1 + p * 1024, dst1 + p * 1024);
                } else {
                    const int p = wave + 4 * (k - 11);
                    glds(src2 + p * 1024, dst2 + p * 1024);
                }
            };
            const char* w1 = smem + (PAR ^ 1) * FF_W1_BYTES;
            const char* w1l = w1 + lane * 16;
            const char* w2l = smem + FF_LDS_W2 + PAR * FF_W2_BYTES + lane * 16;
            uint4 fr[4];
            if constexpr (!LAST) {
                bias_init(w1, sn);
#pragma unroll
                for (int i = 0; i < 3; ++i) fr[i] = *(const uint4*)(w1l + i * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t hp[8];
            svd_f32x2 gt, gmx, gr, gq;
            const svd_f32x2 k6 = {1.775515804e-05f, 1.775515804e-05f}, k5 = {-6.477575890e-04f, -6.477575890e-04f}, k4 = {7.724042874e-03f, 7.724042874e-03f},
                            k3 = {-5.292673725e-02f, -5.292673725e-02f}, k2 = {-4.590827371e-01f, -4.590827371e-01f}, k1 = {-1.151116856e+00f, -1.151116856e+00f},
                            km1 = {-1.0f, -1.0f};
#pragma unroll
            for (int i = 0; i < 2 * FF_NS; ++i) {
                if constexpr (!LAST) {
                    if (i + 3 < 2 * FF_NS) fr[(i + 3) & 3] = *(const uint4*)(w1l + (i + 3) * 1024);
                }
                if constexpr (PV & 2) { if (i % 5 == 4) hp[i / 5] = __builtin_bit_cast(uint32_t, sc[(i / 5) >> 2][2 * ((i / 5) & 3)]); } else
                {   // GELU pair p = i / 5 (tile p / 4, registers 2 (p % 4), + 1), stage i % 5: the arithmetic of gelu_erf_f2 (svd_common.h), bit for bit
                    const int pr = i / 5, st = i % 5, t = pr >> 2, e = 2 * (pr & 3);
                    if (st == 0) {
                        asm("v_min_f32 %0, %2, |%1|" : "=v"(gt[0]) : "v"(sc[t][8 + e]), "s"(7.0f));
                        asm("v_min_f32 %0, %2, |%1|" : "=v"(gt[1]) : "v"(sc[t][9 + e]), "s"(7.0f));
                        asm("v_max_f32 %0, 0, %1" : "=v"(gmx[0]) : "v"(sc[t][8 + e]));
                        asm("v_max_f32 %0, 0, %1" : "=v"(gmx[1]) : "v"(sc[t][9 + e]));
                        gr = __builtin_elementwise_fma(k6, gt, k5);
                    } else if (st == 1) {
                        gr = __builtin_elementwise_fma(gr, gt, k4);
                        gr = __builtin_elementwise_fma(gr, gt, k3);
                    } else if (st == 2) {
                        gr = __builtin_elementwise_fma(gr, gt, k2);
                        gr = __builtin_elementwise_fma(gr, gt, k1);
                    } else if (st == 3) {
                        const svd_f32x2 ex = __builtin_elementwise_fma(gr, gt, km1);
                        gq[0] = __builtin_amdgcn_exp2f(ex[0]); gq[1] = __builtin_amdgcn_exp2f(ex[1]);
                    } else {
                        svd_f32x2 ax;
                        ax[0] = __builtin_fabsf(sc[t][8 + e]); ax[1] = __builtin_fabsf(sc[t][9 + e]);
                        const svd_f32x2 gl = __builtin_elementwise_fma(-ax, gq, gmx);
                        hp[pr] = E::pack(sc[t][e] * gl[0], sc[t][e + 1] * gl[1]);
                    }
                }
                if constexpr (!(PV & 1)) {                                    // all 16 pieces in phase A (slots 1, 3, 6, 8, .., 36, 38): they must be OLDER than
                    if (i % 5 == 1) dma(2 * (i / 5));                         // the row loads of phase B for the vmcnt(20) below to cover them
                    if (i % 5 == 3) dma(2 * (i / 5) + 1);
                }
                if constexpr (!LAST && PV != 4) sn[i & 1] = E::mfma(fr[i & 3], xf[i >> 1], sn[i & 1]);
                if constexpr (!LAST && PV == 4) sn[i & 1][0] += __builtin_bit_cast(float, fr[i & 3].x);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint4 hf[2];
            hf[0].x = hp[0]; hf[0].y = hp[1]; hf[0].z = hp[2]; hf[0].w = hp[3];
            hf[1].x = hp[4]; hf[1].y = hp[5]; hf[1].z = hp[6]; hf[1].w = hp[7];
            int xrow = 0;
            if constexpr (LOADX) {           // xf is dead after the last S^T of the tile: request the next tile's rows under the rest of this one
                const int nt = tile + (int)gridDim.x;
                xrow = (nt < ntiles ? nt : tile) * 128 + wave * 32 + l31;
                xrow = xrow < M ? xrow : M - 1;
            }
            const svd_bf16* xp = X + (int64_t)xrow * ldx + 8 * hi;
            uint4 f2[4];
#pragma unroll
            for (int j = 0; j < 3; ++j) f2[j] = *(const uint4*)(w2l + j * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2 * FF_NO; ++j) {
                if (j + 3 < 2 * FF_NO) f2[(j + 3) & 3] = *(const uint4*)(w2l + (j + 3) * 1024);
                if constexpr (LOADX) xf[j] = *(const uint4*)(xp + 16 * j);
                o_acc[j % FF_NO] = E::mfma(f2[j & 3], hf[j / FF_NO], o_acc[j % FF_NO]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (LOADX) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // the 16 DMA pieces of this step; the 20 row loads stay in flight
            else svd_wait_dma();
            __syncthreads();
        };
        for (int c = 0; c + 2 < nch; c += 2) {
            step.template operator()<0, false, false>(c, s_a, s_b);
            step.template operator()<1, false, false>(c + 1, s_b, s_a);
        }
        step.template operator()<0, false, true>(nch - 2, s_a, s_b);
        step.template operator()<1, true, false>(nch - 1, s_b, s_a);
        // ---- epilogue: lane holds row l31, channels 32 o + 8 j + 4 hi .. + 3 in o_acc[o][4 j .. 4 j + 3]
        const int row = (tile * 128 + wave * 32 + l31);
        if (row < M) {
            const float* bl = (const float*)(smem + FF_LDS_B2) + 4 * hi;
#pragma unroll
            for (int o = 0; o < FF_NO; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = 32 * o + 8 * j;
                    const float4 b = *(const float4*)(bl + ch);
                    float4 v = {o_acc[o][4 * j] + b.x, o_acc[o][4 * j + 1] + b.y, o_acc[o][4 * j + 2] + b.z, o_acc[o][4 * j + 3] + b.w};
                    if constexpr (RES == 2) {
                        const float4 r = *(const float4*)((const float*)R + (int64_t)row * ldr + ch + 4 * hi);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    } else if constexpr (RES == 1) {
                        const uint2 r = *(const uint2*)((const svd_bf16*)R + (int64_t)row * ldr + ch + 4 * hi);
                        v.x += E::lo(r.x); v.y += E::hi(r.x); v.z += E::lo(r.y); v.w += E::hi(r.y);
                    }
                    if constexpr (BLEND) {
                        const float beta = 1.0f - alpha;
                        float4 sv;
                        if constexpr (RES == 2) {
                            sv = *(const float4*)((const float*)S + (int64_t)row * lds + ch + 4 * hi);
                        } else {
                            const uint2 r = *(const uint2*)((const svd_bf16*)S + (int64_t)row * lds + ch + 4 * hi);
                            sv.x = E::lo(r.x); sv.y = E::hi(r.x); sv.z = E::lo(r.y); sv.w = E::hi(r.y);
                        }
                        v.x = alpha * sv.x + beta * v.x; v.y = alpha * sv.y + beta * v.y; v.z = alpha * sv.z + beta * v.z; v.w = alpha * sv.w + beta * v.w;
                    }
                    if constexpr (OUT32) {
                        *(float4*)((float*)Y + (int64_t)row * ldy + ch + 4 * hi) = v;
                    } else {
                        uint2 u; u.x = E::pack(v.x, v.y); u.y = E::pack(v.z, v.w);
                        *(uint2*)((svd_bf16*)Y + (int64_t)row * ldy + ch + 4 * hi) = u;
                    }
                }
        }
    }
    svd_wait_dma();          // the ring's last requests (chunks of a tile that does not exist) must not outlive the workgroup's LDS
}

}  // namespace

// X [M, 320] 16-bit rows (the LayerNorm output), Wp: packed image (svd_ff_fused_pack_bytes(hidden) bytes, layout above / ops.pack_ff_fused),
// b2 [320] fp32, R: residual rows (fp32 when res_f32, else 16 bit; may be NULL), Y: fp32 rows when out_f32, else 16 bit.  hidden % 64 == 0.
extern "C" int64_t svd_ff_fused_pack_bytes(int32_t hidden) { return hidden > 0 && hidden % 64 == 0 ? (int64_t)(hidden / 32) * FF_BLOB : (int64_t)SVD_EINVAL; }

#ifdef SVD_FF_PROBES
extern "C" { int svd_ff_probe_variant = 0; }
#endif
extern "C" int svd_ff_geglu_fused(const svd_bf16* X, int64_t ldx, const void* Wp, int32_t channels, int32_t hidden, const float* b2, const void* R,
                                  int64_t ldr, const void* S, int64_t lds, float alpha, int32_t res_f32, void* Y, int64_t ldy, int32_t out_f32, int64_t M,
                                  int32_t dtype, svd_stream_t stream) {
    if (!X || !Wp || !b2 || !Y || M <= 0 || M > 0x7fffff00 || channels != FF_C || hidden <= 0 || hidden % 64) return SVD_EINVAL;
    if (ldx % 8 || ldx < FF_C || ldy < FF_C || (R && ldr < FF_C)) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wp) & 15) return SVD_EINVAL;
    if (out_f32 ? ((ldy % 4) || ((uintptr_t)Y & 15)) : ((ldy % 4) || ((uintptr_t)Y & 7))) return SVD_EINVAL;
    if (R && (res_f32 ? ((ldr % 4) || ((uintptr_t)R & 15)) : ((ldr % 4) || ((uintptr_t)R & 7)))) return SVD_EINVAL;
    if (S && (!R || lds < FF_C || (res_f32 ? ((lds % 4) || ((uintptr_t)S & 15)) : ((lds % 4) || ((uintptr_t)S & 7))))) return SVD_EINVAL;   // blend: only with a residual
    const int ntiles = (int)((M + 127) / 128);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return SVD_ELAUNCH;
        n_cu = p.multiProcessorCount;
    }
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const int nch = hidden / 32;
#define FF_LAUNCH(RES, OUT)  FF_LAUNCH2(RES, OUT, false)
#define FF_LAUNCH2(RES, OUT, BL)                                                                                                            \
    do {                                                                                                                                 \
        SVD_DISPATCH_DTYPE(dtype, {                                                                                                      \
            auto kern = ff_geglu_fused_kernel<E, RES, OUT, BL>;                                                                           \
            static bool attr_set = false;                                                                                                \
            if (!attr_set) {                                                                                                             \
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
                attr_set = true;                                                                                                         \
            }                                                                                                                            \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), FF_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, \
                               alpha, Y, ldy, (int)M, ntiles);                                                                                          \
        });                                                                                                                              \
    } while (0)
#ifdef SVD_FF_PROBES
    if (svd_ff_probe_variant) {
#define FF_PROBE(PVV)                                                                                                                    \
        do {                                                                                                                             \
            auto kern = ff_geglu_fused_kernel<ElemF16, 2, true, false, PVV>;                                                                   \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), FF_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, alpha, Y, ldy, (int)M, ntiles); \
        } while (0)
        switch (svd_ff_probe_variant) { case 1: FF_PROBE(1); break; case 2: FF_PROBE(2); break; case 3: FF_PROBE(3); break; default: FF_PROBE(4); break; }
        SVD_CHECK_LAUNCH("ff_geglu_fused(probe)");
        return SVD_OK;
    }
#endif
    const int res = R ? (res_f32 ? 2 : 1) : 0;
    if (S) {
        if (res == 2) { if (out_f32) FF_LAUNCH2(2, true, true); else FF_LAUNCH2(2, false, true); }
        else { if (out_f32) FF_LAUNCH2(1, true, true); else FF_LAUNCH2(1, false, true); }
    } else if (res == 2) { if (out_f32) FF_LAUNCH(2, true); else FF_LAUNCH(2, false); }
    else if (res == 1) { if (out_f32) FF_LAUNCH(1, true); else FF_LAUNCH(1, false); }
    else { if (out_f32) FF_LAUNCH(0, true); else FF_LAUNCH(0, false); }
#undef FF_LAUNCH
#undef FF_LAUNCH2
    SVD_CHECK_LAUNCH("ff_geglu_fused");
    return SVD_OK;
}
