// Fused GEGLU feed-forward for the 320-channel level (round 5):   Y = R + b2 + W2 . ( (W1v X + b1v) * gelu_erf(W1g X + b1g) )
//
// Replaces the TWO launches of FeedForward (code/models/svd/sgm/modules/attention.py:94-120: GEGLU.proj + gate, then net[2]; diffusers
// FeedForward(activation_fn="geglu") in the enhancer, code/i2v_enhance/attention.py:414-534) for dim = 320, inner = 1280 -- the level-0 blocks
// of the VideoUNet / ControlNet / I2VGenXLUNet, where the [M, 1280] hidden tensor is the largest byte-mover of a forward (1.18 GB written by
// the projection and read straight back by the down-projection at M = 460 800).  Here the hidden tile never leaves the CU.  Two bit-identical forms of
// the kernel live in this file: ff_geglu_fused_kernel (four waves, one per SIMD: described first) and ff_geglu_fused8_kernel (eight waves: wave PAIRS
// share the 32 rows and split the hidden tiles / output channels, described at its definition) -- the eight-wave form is the default since the epilogue's
// residual loads are batched (1.24 vs 1.31 ms at M = 460 800; SVD_FF_WAVES=4 selects the other).  Common to both:
//
//   * one WAVE (pair) owns 32 token rows for the whole computation; 4 waves = one 128-row workgroup tile, one wave per SIMD (the accumulators of
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
#include <stdlib.h>

namespace {

#ifdef SVD_FF_PROBES
__device__ int ff_dephase_probe = -1;          // probe builds: >= 0 overrides FF_DEPHASE_SLEEPS (units of s_sleep 127 per start slot)
#endif
constexpr int FF_DEPHASE_SLEEPS = 0;             // start de-phasing of the workgroups (see ff_start_delay); 0 = off: MEASURED, no gain (profiles/r05_ff_fused_probe_dephase.txt:
                                                 // 1.382 ms -> 1.351 / 1.391 / 1.435 ms with slots of 4 / 8 / 12 us at M = 460 800; slower at every setting at M = 129 024)

// All workgroups walk the same number of equal tiles: without help every CU runs its tile prologue / epilogue (X, R in, Y out) at the same moment and
// its MFMA phases at the same moment -- HBM idles, then saturates.  Workgroup b starts (b mod 16) slots late; a slot is `sleeps` x s_sleep 127 (~4 us).
__device__ __forceinline__ void ff_start_delay() {
    int n = FF_DEPHASE_SLEEPS;
#ifdef SVD_FF_PROBES
    if (ff_dephase_probe >= 0) n = ff_dephase_probe;
#endif
    const int k = (blockIdx.x & 15) * n;
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
}

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

// Epilogue of both forms (round 6): acc[o][r] = O[row rbase + (r & 3) + 8 (r >> 2)][channel ch0 + 32 o] of this lane; + b2 (LDS, bl[32 o]), + residual, blend, store.
// One dword (or 16-bit) access per register: 2 token rows x 128 (64) contiguous bytes per instruction.  The residual (and blend partner) loads of 4 accumulator
// registers x NO tiles are issued together before their first use (the S^T accumulators and fragment rings are dead here).
// LN: also Yn = LayerNorm(V + addvec[frame]) * gamma + beta in the 16-bit type (the nn.LayerNorm that consumes the block's result: norm_in of the time_stack with
// x + time_pos_embed, norm1 behind ff_in; video_attention.py:125-168,318-321): the 32 lanes of a half-wave hold the same 16 tokens, so the statistics are a 32-lane
// butterfly per token over this wave's NO x 32 channels; in the eight-wave form (PAIRX) the two waves of a pair hold one half of a token's 320 channels each and merge
// (mean, M2) of their halves through LDS (Chan's parallel update: exact, one exchange, one workgroup barrier).
#ifndef SVD_FF_LN_PROBE
#define SVD_FF_LN_PROBE 0          // developer builds only (cost attribution of the LayerNorm epilogue): 1 no Yn stores, 2 no wave-pair exchange / barrier, 4 no reductions
#endif
struct FfLn { const float* gamma; const float* beta; float eps; const float* addvec; svd_bf16* Yn; int64_t ldyn; float* xch_own; const float* xch_other; };
template <class E, int RES, bool OUT32, bool BLEND, int NO, bool LN, bool PAIRX>
__device__ __forceinline__ void ff_epilogue(f32x16_t (&acc)[NO], int rbase, int M, int ch0, int l31, int hi, const float* bl, const float* vecp, const void* __restrict__ R,
                                            int64_t ldr, const void* __restrict__ S, int64_t lds, float alpha, void* __restrict__ Y, int64_t ldy, const FfLn& ln) {
    float bias[NO];          // b2 + the per-frame vector of this wave's 32 rows (vecp: already at the frame's row and this lane's channel; NULL = none)
#pragma unroll
    for (int o = 0; o < NO; ++o) bias[o] = bl[32 * o] + (vecp ? vecp[32 * o] : 0.f);
    // LN operands: loaded HERE, ahead of the residual loads -- issued behind the Y stores they would wait for every one of them (vmcnt counts in order)
    float gam[LN ? NO : 1], bet[LN ? NO : 1], av[LN ? NO : 1];
    if constexpr (LN) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            gam[o] = ln.gamma[ch0 + 32 * o]; bet[o] = ln.beta[ch0 + 32 * o];
            av[o] = ln.addvec ? ln.addvec[32 * o] : 0.f;          // addvec: already at the frame's row and this lane's channel
        }
    }
    constexpr int G = 4;                                       // registers (token rows) per batch
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += G) {
        int64_t trow[G];
        float rv[G][NO], sv[BLEND ? G : 1][BLEND ? NO : 1];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = r0 + g;
            const int t = rbase + (r & 3) + 8 * (r >> 2);
            trow[g] = t < M ? t : M - 1;
            if constexpr (RES == 2) {
                const float* rp = (const float*)R + trow[g] * ldr + ch0;
#pragma unroll
                for (int o = 0; o < NO; ++o) rv[g][o] = rp[32 * o];
                if constexpr (BLEND) {
                    const float* sp = (const float*)S + trow[g] * lds + ch0;
#pragma unroll
                    for (int o = 0; o < NO; ++o) sv[g][o] = sp[32 * o];
                }
            } else if constexpr (RES == 1) {
                const svd_bf16* rp = (const svd_bf16*)R + trow[g] * ldr + ch0;
#pragma unroll
                for (int o = 0; o < NO; ++o) rv[g][o] = E::to_f32(rp[32 * o]);
                if constexpr (BLEND) {
                    const svd_bf16* sp = (const svd_bf16*)S + trow[g] * lds + ch0;
#pragma unroll
                    for (int o = 0; o < NO; ++o) sv[g][o] = E::to_f32(sp[32 * o]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int r = r0 + g;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                float v = acc[o][r] + bias[o];
                if constexpr (RES != 0) v += rv[g][o];
                if constexpr (BLEND) v = alpha * sv[g][o] + (1.0f - alpha) * v;
                if constexpr (OUT32) ((float*)Y)[trow[g] * ldy + ch0 + 32 * o] = v;
                else ((svd_bf16*)Y)[trow[g] * ldy + ch0 + 32 * o] = E::from_f32(v);
                if constexpr (LN) acc[o][r] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LN) {
#pragma unroll
        for (int o = 0; o < NO; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][r] += av[o];
        constexpr float inv_w = 1.f / (float)(32 * NO);
        float mean[16], m2[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sm = 0.f;
#pragma unroll
            for (int o = 0; o < NO; ++o) sm += acc[o][r];
            mean[r] = ((SVD_FF_LN_PROBE & 4) ? sm : half_wave_sum(sm)) * inv_w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sq = 0.f;
#pragma unroll
            for (int o = 0; o < NO; ++o) { const float d = acc[o][r] - mean[r]; sq += d * d; }
            m2[r] = (SVD_FF_LN_PROBE & 4) ? sq : half_wave_sum(sq);
        }
        if constexpr (PAIRX && !(SVD_FF_LN_PROBE & 2)) {
            if (l31 == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) *(float4*)(ln.xch_own + hi * 32 + 2 * r) = make_float4(mean[r], m2[r], mean[r + 1], m2[r + 1]);
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float4 p = *(const float4*)(ln.xch_other + hi * 32 + 2 * r);
                const float d0 = mean[r] - p.x, d1 = mean[r + 1] - p.z;
                m2[r] += p.y + (float)(16 * NO) * d0 * d0;          // n_a n_b / (n_a + n_b) = (32 NO)^2 / (64 NO)
                m2[r + 1] += p.w + (float)(16 * NO) * d1 * d1;
                mean[r] = 0.5f * (mean[r] + p.x);
                mean[r + 1] = 0.5f * (mean[r + 1] + p.z);
            }
        }
        constexpr float inv_c = 1.f / (float)FF_C;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float rstd = rsqrtf(m2[r] * inv_c + ln.eps);
            const int t = rbase + (r & 3) + 8 * (r >> 2);
            svd_bf16* np = ln.Yn + (int64_t)(t < M ? t : M - 1) * ln.ldyn + ch0;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const svd_bf16 q = E::from_f32((acc[o][r] - mean[r]) * rstd * gam[o] + bet[o]);
                if constexpr (SVD_FF_LN_PROBE & 1) asm volatile("" :: "v"(q));
                else np[32 * o] = q;
            }
        }
    }
}

// RES: 0 = no residual, 1 = 16-bit residual rows, 2 = fp32 residual rows (the fp32 residual stream); OUT32: fp32 output rows;
// BLEND: Y = alpha * S + (1 - alpha) * (...) with S of the residual's type (the temporal block's AlphaBlender, video_attention.py:318-322)
// PV (probe builds only, -DSVD_FF_PROBES; results are WRONG for PV != 0): 1 = no LDS-DMA in the steps (stale weights), 2 = no GELU arithmetic,
// 3 = neither, 4 = no MFMA of S^T (phase A: reads + GELU + DMA only)
template <class E, int RES, bool OUT32, bool BLEND = false, int PV = 0, bool LN = false>
__global__ __launch_bounds__(256, 1) void ff_geglu_fused_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const char* __restrict__ Wp, int nch,
                                                                const float* __restrict__ b2, const void* __restrict__ R, int64_t ldr,
                                                                const void* __restrict__ S, int64_t lds, float alpha,
                                                                void* __restrict__ Y, int64_t ldy, int M, int ntiles,
                                                                const float* __restrict__ rowvec, int rowvec_ld, int rows_per_vec,
                                                                const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps,
                                                                const float* __restrict__ ln_addvec, int ln_addvec_ld, int ln_rows_per_vec,
                                                                svd_bf16* __restrict__ Yn, int64_t ldyn, int stagger_ticks, int stagger_from) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ff_start_delay();
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
    // PHASE STAGGER (round 6): the workgroups of a launch start together and every tile takes the same time, so all 256 CUs reach their epilogue -- the part of a tile
    // that moves its HBM bytes (residual in, rows out) -- at the same moment, then leave the memory system idle through the next tile's MFMA phases.  The workgroups from
    // `stagger_from` on start `stagger_ticks` (100 MHz wall clock; half a tile) late: two groups in anti-phase.  The host picks them among the workgroups that own one
    // tile less than the others, so the delay hides in the tail of the launch.  Measured 1-2 % of the kernel (profiles/r06_ff_stagger_ab.txt).
    if (stagger_ticks > 0 && (int)blockIdx.x >= stagger_from) {
        const long long until = wall_clock64() + stagger_ticks;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(64);
    }
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
                o_acc[j % FF_NO] = E::mfma(hf[j / FF_NO], f2[j & 3], o_acc[j % FF_NO]);       // O[row, channel]: H is the A operand (round 6: see the epilogue)
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
        // ---- epilogue (round 6: untransposed).  The packed GEGLU fragment has the same register image as an A operand (lane = token l31, k = 8 hi + e) and the
        // W2 fragment as a B operand (lane = channel, k = 8 hi + e), so GEMM 2 is issued as O = H . W2^T: o_acc[o][r] = O[token (r & 3) + 8 (r >> 2) + 4 hi]
        // [channel 32 o + l31] -- A LANE IS A CHANNEL, and one dword load / store moves 2 token rows x 128 contiguous bytes per instruction instead of 32 rows x 32 B
        // (tools/access_pattern_bench.hip: 5.3 against 2.7 TB/s for this kernel's R-in / Y-out traffic as a plain copy at four waves per CU).  Rows past M were
        // loaded from row M - 1 (load_x clamps) and recompute ITS values: they store duplicates to row M - 1 (no exec-mask branch per store).
        {
            const int rbase = tile * 128 + wave * 32 + 4 * hi;
            const float* bl = (const float*)(smem + FF_LDS_B2) + l31;
            // rows_per_vec % 32 == 0: one vector per wave; a wave wholly past M recomputes row M - 1 and must take ITS vector (its stores are duplicates of that row)
            const int vrow = tile * 128 + wave * 32 < M ? tile * 128 + wave * 32 : M - 1;
            const float* rv = rowvec ? rowvec + (int64_t)(vrow / rows_per_vec) * rowvec_ld + l31 : nullptr;
            FfLn ln{};
            if constexpr (LN) ln = FfLn{ln_gamma, ln_beta, ln_eps, ln_addvec ? ln_addvec + (int64_t)(vrow / ln_rows_per_vec) * ln_addvec_ld + l31 : nullptr, Yn, ldyn, nullptr, nullptr};
            ff_epilogue<E, RES, OUT32, BLEND, FF_NO, LN, false>(o_acc, rbase, M, l31, l31, hi, bl, rv, R, ldr, S, lds, alpha, Y, ldy, ln);
        }
    }
    svd_wait_dma();          // the ring's last requests (chunks of a tile that does not exist) must not outlive the workgroup's LDS
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Second form (round 5): the same computation on EIGHT waves per workgroup, two per SIMD -- built on the hypothesis that the four-wave kernel above is
// bound by ONE wave's issue stream (~535 instructions per 60 MFMAs against the ~5 fillers per 32-cycle MFMA gap that MI355X_MICROARCH.md measures as free
// for one wave per SIMD).  MEASURED: bit-identical and, as first built, NOT faster (1.39 vs 1.38 ms at M = 460 800) -- halving the per-wave stream does not
// shorten the step.  What the probe variants of both forms say instead (profiles/r05_ff_fused_probe_8waves.txt): the matrix pipe alone would need 0.56 ms
// (registers-only MFMA loop: 2.0-2.07 PFLOP/s at the clock the chip holds); with DMA, GELU, barrier AND fragment reads stripped the eight-wave kernel still
// took 0.83 ms -- the remainder was the tile epilogue, whose residual loads the compiler had serialised into one HBM round trip per fragment (batched since:
// see the epilogue) -- and the four removed pieces cost 0.16 (LDS-DMA) / 0.09 (GELU) / 0.11 (barrier) / 0.20 ms (fragment reads) when removed one at a time.
// Deeper fragment prefetch (7 ahead) changes nothing; a start de-phasing of the workgroups changes nothing.  With the batched epilogue this form is 5 % ahead of
// the four-wave one (1.24 vs 1.31 ms) and is the default (SVD_FF_WAVES=4 selects the other).  Here a PAIR of waves shares 32 token rows: both hold the rows' X fragments; wave q of the pair computes
// S^T of hidden tile q of every chunk (20 MFMAs), gates it, and hands its packed 16-unit GEGLU fragment to the partner through a 1-KiB LDS slot
// (written before the step's workgroup barrier, read after it); each wave then accumulates ITS half of the output channels (5 tiles of 32) over both
// hidden tiles (10 MFMAs).  Per wave and chunk: 30 MFMAs, 30 fragment reads, 4 GELU pairs, 9 DMA pieces -- half of the four-wave kernel's stream --
// and a SIMD always has a second wave to issue from.  Same packed weight image, same arithmetic per element in the same order (GEMM 2 adds hidden
// tile 0 before tile 1 in both forms), so the results are bit-identical to the four-wave kernel.
constexpr int F8_LDS_X = FF_LDS_TOTAL;                         // exchange: 2 (step parity) x 8 (wave) x 1 KiB
constexpr int F8_LDS_TOTAL = F8_LDS_X + 2 * 8 * 1024;          // 142 592 B
constexpr int F8_NO = FF_NO / 2;                               // 5 output tiles (160 channels) per wave

// PV (probe builds only; results are WRONG for PV != 0; bit mask): 1 = no LDS-DMA in the steps, 2 = no GELU arithmetic, 4 = no workgroup barrier in the steps,
// 8 = no fragment reads in the steps (stale registers), 16 = fragments 7 MFMAs ahead (ring of 8) instead of 3
template <class E, int RES, bool OUT32, bool BLEND = false, int PV = 0, bool LN = false>
__global__ __launch_bounds__(512, 1) void ff_geglu_fused8_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const char* __restrict__ Wp, int nch,
                                                                 const float* __restrict__ b2, const void* __restrict__ R, int64_t ldr,
                                                                 const void* __restrict__ S, int64_t lds, float alpha,
                                                                 void* __restrict__ Y, int64_t ldy, int M, int ntiles,
                                                                 const float* __restrict__ rowvec, int rowvec_ld, int rows_per_vec,
                                                                 const float* __restrict__ ln_gamma, const float* __restrict__ ln_beta, float ln_eps,
                                                                 const float* __restrict__ ln_addvec, int ln_addvec_ld, int ln_rows_per_vec,
                                                                 svd_bf16* __restrict__ Yn, int64_t ldyn, int stagger_ticks, int stagger_from) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, q = wave & 1;
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t voff = (uint32_t)lane * 16u;

    auto glds = [&](const char* src, uint32_t dst) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(dst) : "m0");
    };
    // piece k of this wave's 6 W1 pieces (wave + 8 k, clamped to the last piece: the clamped copies write the same bytes to the same place) / 3 W2 pieces
    auto dma_w1_piece = [&](const char* src, uint32_t dst, int k) __attribute__((always_inline)) {
        int p = wave + 8 * k;
        p = p > FF_W1_PIECES - 1 ? FF_W1_PIECES - 1 : p;
        glds(src + p * 1024, dst + p * 1024);
    };
    auto dma_w2_piece = [&](const char* src, uint32_t dst, int k) __attribute__((always_inline)) {
        int p = wave + 8 * k;
        p = p > FF_W2_PIECES - 1 ? FF_W2_PIECES - 1 : p;
        glds(src + p * 1024, dst + p * 1024);
    };

    uint4 xf[FF_NS];
    auto load_x = [&](int tile) {
        int row = tile * 128 + pair * 32 + l31;
        row = row < M ? row : M - 1;
        const svd_bf16* xp = X + (int64_t)row * ldx + 8 * hi;
#pragma unroll
        for (int s = 0; s < FF_NS; ++s) xf[s] = *(const uint4*)(xp + 16 * s);
    };
    // S^T of THIS wave's tile (q) starts from its 32 biases: rows 8 j + 4 hi + e of the tile <-> registers 4 j + e
    auto bias_init = [&](const char* w, f32x16_t& s_acc) __attribute__((always_inline)) {
        const char* bp = w + 2 * FF_NS * 1024 + q * 128 + hi * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 b = *(const float4*)(bp + j * 32);
            s_acc[4 * j + 0] = b.x; s_acc[4 * j + 1] = b.y; s_acc[4 * j + 2] = b.z; s_acc[4 * j + 3] = b.w;
        }
    };
    auto gemm1_plain = [&](int slot, f32x16_t& s_acc) __attribute__((always_inline)) {
        const char* w = smem + slot * FF_W1_BYTES;
        bias_init(w, s_acc);
        const char* wl = w + q * 1024 + lane * 16;             // fragment (k-step s, tile q) is piece 2 s + q
        uint4 fr[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) fr[i] = *(const uint4*)(wl + i * 2048);
#pragma unroll
        for (int i = 0; i < FF_NS; ++i) {
            if (i + 3 < FF_NS) fr[(i + 3) & 3] = *(const uint4*)(wl + (i + 3) * 2048);
            s_acc = E::mfma(fr[i & 3], xf[i], s_acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    f32x16_t o_acc[F8_NO];

    for (int i = tid; i < FF_C; i += 512) ((float*)(smem + FF_LDS_B2))[i] = b2[i];
    int tile = blockIdx.x;
    if (tile < ntiles) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dma_w1_piece(Wp, sbase, k);
#pragma unroll
        for (int k = 0; k < 6; ++k) dma_w1_piece(Wp + (int64_t)(1 % nch) * FF_BLOB, sbase + FF_W1_BYTES, k);
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_w2_piece(Wp + FF_W1_BYTES, sbase + FF_LDS_W2, k);
        load_x(tile);
    }
    svd_wait_dma();
    __syncthreads();

    // PHASE STAGGER (round 6): the workgroups of a launch start together and every tile takes the same time, so all 256 CUs reach their epilogue -- the part of a tile
    // that moves its HBM bytes (residual in, rows out) -- at the same moment, then leave the memory system idle through the next tile's MFMA phases.  The workgroups from
    // `stagger_from` on start `stagger_ticks` (100 MHz wall clock; half a tile) late: two groups in anti-phase.  The host picks them among the workgroups that own one
    // tile less than the others, so the delay hides in the tail of the launch.  Measured 1-2 % of the kernel (profiles/r06_ff_stagger_ab.txt).
    if (stagger_ticks > 0 && (int)blockIdx.x >= stagger_from) {
        const long long until = wall_clock64() + stagger_ticks;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(64);
    }
    f32x16_t s_a, s_b;
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int o = 0; o < F8_NO; ++o)
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[o][i] = 0.f;
        gemm1_plain(0, s_a);                 // S^T tile q of chunk 0 (W1 slot 0: nch is even, every tile starts on slot parity 0)
        __syncthreads();                     // every wave is done with W1 slot 0 before step 0 overwrites it with chunk 2
        // One chunk.  `sc` holds S^T (tile q) of chunk c, `sn` receives that of chunk c + 1 from W1 slot PAR ^ 1.
        //   phase A, 20 slots: MFMA i of S^T(c + 1) | the fragment read for MFMA i + 3 | one fifth of a GELU pair of chunk c (4 pairs x 5 stages) |
        //                      the wave's 6 W1 pieces of chunk c + 2 (into the slot chunk c was read from one step ago) at slots 1, 4, 7, 10, 13, 16
        //   hand-over: the packed GEGLU fragment -> LDS, wait for the DMA, workgroup barrier, the partner's fragment <- LDS
        //   phase B, 10 slots: MFMA of O^T(c) (hidden tile 0, then tile 1: one of them is the partner's) | the W2 fragment read 3 ahead | the wave's 3 W2 pieces of
        //                      chunk c + 1 at slots 1, 4, 7 (only now: the slot they overwrite was read in phase B of the step before, which every
        //                      wave has left once it passed this step's barrier) | (LOADX) two row loads of the next tile per slot
        auto step = [&]<int PAR, bool LAST, bool LOADX>(int c, f32x16_t& sc, f32x16_t& sn) __attribute__((always_inline)) {
            int c2 = c + 2; c2 = c2 >= nch ? c2 - nch : c2;
            int c1 = c + 1; c1 = c1 >= nch ? 0 : c1;
            const char* src1 = Wp + (int64_t)c2 * FF_BLOB;
            const char* src2 = Wp + (int64_t)c1 * FF_BLOB + FF_W1_BYTES;
            const uint32_t dst1 = sbase + PAR * FF_W1_BYTES, dst2 = sbase + FF_LDS_W2 + (PAR ^ 1) * FF_W2_BYTES;
            const char* w1 = smem + (PAR ^ 1) * FF_W1_BYTES;
            const char* w1l = w1 + q * 1024 + lane * 16;
            constexpr int RD = (PV & 16) ? 8 : 4;            // fragment ring: RD - 1 MFMAs ahead
            uint4 fr[RD];
            if constexpr (!LAST) {
                bias_init(w1, sn);
#pragma unroll
                for (int i = 0; i < RD - 1; ++i) fr[i] = *(const uint4*)(w1l + i * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
            uint32_t hp[4];
            svd_f32x2 gt, gmx, gr, gq;
            const svd_f32x2 k6 = {1.775515804e-05f, 1.775515804e-05f}, k5 = {-6.477575890e-04f, -6.477575890e-04f}, k4 = {7.724042874e-03f, 7.724042874e-03f},
                            k3 = {-5.292673725e-02f, -5.292673725e-02f}, k2 = {-4.590827371e-01f, -4.590827371e-01f}, k1 = {-1.151116856e+00f, -1.151116856e+00f},
                            km1 = {-1.0f, -1.0f};
#pragma unroll
            for (int i = 0; i < FF_NS; ++i) {
                if constexpr (!LAST && !(PV & 8)) {
                    if (i + RD - 1 < FF_NS) fr[(i + RD - 1) & (RD - 1)] = *(const uint4*)(w1l + (i + RD - 1) * 2048);
                }
                if constexpr (PV & 2) { if (i % 5 == 4) hp[i / 5] = __builtin_bit_cast(uint32_t, sc[2 * (i / 5)]); } else
                {   // GELU pair pr = i / 5 (registers 2 pr, 2 pr + 1: value; 8 + 2 pr, 9 + 2 pr: gate), stage i % 5: gelu_erf_f2 (svd_common.h), bit for bit
                    const int pr = i / 5, st = i % 5, e = 2 * pr;
                    if (st == 0) {
                        asm("v_min_f32 %0, %2, |%1|" : "=v"(gt[0]) : "v"(sc[8 + e]), "s"(7.0f));
                        asm("v_min_f32 %0, %2, |%1|" : "=v"(gt[1]) : "v"(sc[9 + e]), "s"(7.0f));
                        asm("v_max_f32 %0, 0, %1" : "=v"(gmx[0]) : "v"(sc[8 + e]));
                        asm("v_max_f32 %0, 0, %1" : "=v"(gmx[1]) : "v"(sc[9 + e]));
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
                        ax[0] = __builtin_fabsf(sc[8 + e]); ax[1] = __builtin_fabsf(sc[9 + e]);
                        const svd_f32x2 gl = __builtin_elementwise_fma(-ax, gq, gmx);
                        hp[pr] = E::pack(sc[e] * gl[0], sc[e + 1] * gl[1]);
                    }
                }
                if constexpr (!(PV & 1)) { if (i % 3 == 1 && i / 3 < 6) dma_w1_piece(src1, dst1, i / 3); }
                if constexpr (!LAST) sn = E::mfma(fr[i & (RD - 1)], xf[i], sn);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- hand-over of the GEGLU fragments inside the pair, and the step's barrier
            uint4 own;
            own.x = hp[0]; own.y = hp[1]; own.z = hp[2]; own.w = hp[3];
            *(uint4*)(smem + F8_LDS_X + ((PAR * 8 + wave) << 10) + lane * 16) = own;
            svd_wait_dma();
            if constexpr (!(PV & 4)) __syncthreads();
            const uint4 other = *(const uint4*)(smem + F8_LDS_X + ((PAR * 8 + (wave ^ 1)) << 10) + lane * 16);
            // ---- O^T(c): fragment (tile t, output tile o) is piece t * 10 + o of the W2 slot; this wave: o = 5 q .. 5 q + 4, t = 0 then t = 1
            const uint4 h0 = q == 0 ? own : other, h1 = q == 0 ? other : own;
            const char* w2l = smem + FF_LDS_W2 + PAR * FF_W2_BYTES + (5 * q) * 1024 + lane * 16;
            auto w2_piece = [&](int j) __attribute__((always_inline)) { return w2l + ((j / F8_NO) * FF_NO + (j % F8_NO)) * 1024; };
            int xrow = 0;
            if constexpr (LOADX) {
                const int nt = tile + (int)gridDim.x;
                xrow = (nt < ntiles ? nt : tile) * 128 + pair * 32 + l31;
                xrow = xrow < M ? xrow : M - 1;
            }
            const svd_bf16* xp = X + (int64_t)xrow * ldx + 8 * hi;
            uint4 f2[RD];
#pragma unroll
            for (int j = 0; j < RD - 1; ++j) f2[j] = *(const uint4*)w2_piece(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2 * F8_NO; ++j) {
                if constexpr (!(PV & 8)) { if (j + RD - 1 < 2 * F8_NO) f2[(j + RD - 1) & (RD - 1)] = *(const uint4*)w2_piece(j + RD - 1); }
                if constexpr (!(PV & 1)) { if (j % 3 == 1 && j / 3 < 3) dma_w2_piece(src2, dst2, j / 3); }
                if constexpr (LOADX) { xf[2 * j] = *(const uint4*)(xp + 16 * (2 * j)); xf[2 * j + 1] = *(const uint4*)(xp + 16 * (2 * j + 1)); }
                o_acc[j % F8_NO] = E::mfma(j < F8_NO ? h0 : h1, f2[j & (RD - 1)], o_acc[j % F8_NO]);       // O[row, channel] (see the four-wave kernel's epilogue)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int c = 0; c + 2 < nch; c += 2) {
            step.template operator()<0, false, false>(c, s_a, s_b);
            step.template operator()<1, false, false>(c + 1, s_b, s_a);
        }
        step.template operator()<0, false, true>(nch - 2, s_a, s_b);
        step.template operator()<1, true, false>(nch - 1, s_b, s_a);
        // ---- epilogue: o_acc[o][r] = O[token (r & 3) + 8 (r >> 2) + 4 hi of the pair's 32][channel 160 q + 32 o + l31] (untransposed: see the four-wave kernel)
        {
            const int rbase = tile * 128 + pair * 32 + 4 * hi;
            const float* bl = (const float*)(smem + FF_LDS_B2) + 160 * q + l31;
            const int vrow = tile * 128 + pair * 32 < M ? tile * 128 + pair * 32 : M - 1;          // (see the four-wave kernel)
            const float* rv = rowvec ? rowvec + (int64_t)(vrow / rows_per_vec) * rowvec_ld + 160 * q + l31 : nullptr;
            FfLn ln{};
            // (mean, M2) exchange inside the wave pair: the step-parity-0 GEGLU exchange slots of the two waves are dead here -- every wave has passed the last step's
            // barrier -- and are next written after the next tile's first barrier
            if constexpr (LN) ln = FfLn{ln_gamma, ln_beta, ln_eps, ln_addvec ? ln_addvec + (int64_t)(vrow / ln_rows_per_vec) * ln_addvec_ld + 160 * q + l31 : nullptr, Yn, ldyn,
                                        (float*)(smem + F8_LDS_X + (wave << 10)), (const float*)(smem + F8_LDS_X + ((wave ^ 1) << 10))};
            ff_epilogue<E, RES, OUT32, BLEND, F8_NO, LN, true>(o_acc, rbase, M, 160 * q + l31, l31, hi, bl, rv, R, ldr, S, lds, alpha, Y, ldy, ln);
        }
    }
    svd_wait_dma();
}

}  // namespace

// X [M, 320] 16-bit rows (the LayerNorm output), Wp: packed image (svd_ff_fused_pack_bytes(hidden) bytes, layout above / ops.pack_ff_fused),
// b2 [320] fp32, R: residual rows (fp32 when res_f32, else 16 bit; may be NULL), Y: fp32 rows when out_f32, else 16 bit.  hidden % 64 == 0.
extern "C" int64_t svd_ff_fused_pack_bytes(int32_t hidden) { return hidden > 0 && hidden % 64 == 0 ? (int64_t)(hidden / 32) * FF_BLOB : (int64_t)SVD_EINVAL; }

#ifdef SVD_FF_PROBES
extern "C" { int svd_ff_probe_variant = 0; }      // 1..4 / 101..116: timing probes of the four- / eight-wave kernel; -1: the four-wave kernel; 0: the eight-wave kernel
#define FF_FORCE4 || svd_ff_probe_variant == -1
#else
#define FF_FORCE4
#endif
// SVD_FF_STAGGER=<us> (default FF_STAGGER_US = half a tile's time; 0 = off): phase stagger of the workgroups, see the kernels
#ifndef FF_STAGGER_US
#define FF_STAGGER_US 45
#endif
static int ff_stagger_ticks() {
    int us = FF_STAGGER_US;
    if (const char* e = getenv("SVD_FF_STAGGER")) us = atoi(e);
    return (us < 0 || us > 1000) ? 0 : us * 100;
}

extern "C" int svd_ff_geglu_fused(const svd_bf16* X, int64_t ldx, const void* Wp, int32_t channels, int32_t hidden, const float* b2, const void* R,
                                  int64_t ldr, const void* S, int64_t lds, float alpha, int32_t res_f32, void* Y, int64_t ldy, int32_t out_f32, int64_t M,
                                  int32_t dtype, const float* rowvec, int32_t rowvec_ld, int32_t rows_per_vec, const float* ln_gamma, const float* ln_beta,
                                  float ln_eps, const float* ln_addvec, int32_t ln_addvec_ld, int32_t ln_rows_per_vec, svd_bf16* Yn, int64_t ldyn,
                                  svd_stream_t stream) {
    if (rowvec && (rows_per_vec <= 0 || rows_per_vec % 32 || rowvec_ld < FF_C)) return SVD_EINVAL;
    // the fused LayerNorm exists for the fp32-stream form (fp32 residual in, fp32 rows out, no blend): the two places a LayerNorm consumes a feed-forward's result
    if (Yn && (!ln_gamma || !ln_beta || !R || !res_f32 || !out_f32 || S || ldyn < FF_C)) return SVD_EINVAL;
    if (Yn && ln_addvec && (ln_rows_per_vec <= 0 || ln_rows_per_vec % 32 || ln_addvec_ld < FF_C)) return SVD_EINVAL;
    if (!X || !Wp || !b2 || !Y || M <= 0 || M > 0x7fffff00 || channels != FF_C || hidden <= 0 || hidden % 64) return SVD_EINVAL;
    if (ldx % 8 || ldx < FF_C || ldy < FF_C || (R && ldr < FF_C)) return SVD_EINVAL;
    if (((uintptr_t)X | (uintptr_t)Wp) & 15) return SVD_EINVAL;
    if (out_f32 ? ((ldy % 4) || ((uintptr_t)Y & 15)) : ((ldy % 4) || ((uintptr_t)Y & 7))) return SVD_EINVAL;
    if (R && (res_f32 ? ((ldr % 4) || ((uintptr_t)R & 15)) : ((ldr % 4) || ((uintptr_t)R & 7)))) return SVD_EINVAL;
    if (S && (!R || lds < FF_C || (res_f32 ? ((lds % 4) || ((uintptr_t)S & 15)) : ((lds % 4) || ((uintptr_t)S & 7))))) return SVD_EINVAL;   // blend: only with a residual
    const int ntiles = (int)((M + 127) / 128);
    // per-DEVICE caches (relaxed atomics: several host threads may launch; a race only repeats an idempotent query / attribute call)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SVD_ELAUNCH;
    static int n_cu_dev[64] = {0};
    int n_cu = __atomic_load_n(&n_cu_dev[dev], __ATOMIC_RELAXED);
    if (!n_cu) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return SVD_ELAUNCH;
        n_cu = p.multiProcessorCount;
        __atomic_store_n(&n_cu_dev[dev], n_cu, __ATOMIC_RELAXED);
    }
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    const int nch = hidden / 32;
    // stagger only where a workgroup walks several tiles and some workgroups own one tile less than others (their delay is then free): the late group is the upper half
    // of the grid, or what of it owns the smaller tile count
    static const int stg_ticks = ff_stagger_ticks();
    const int rem = ntiles % grid;
    const int stagger_from = rem > grid / 2 ? rem : grid / 2;
    const int stagger_ticks = (ntiles >= 4 * n_cu && rem != 0) ? stg_ticks : 0;
    // SVD_FF_WAVES=4 (A/B switch): the four-wave form (one wave per SIMD) instead of the eight-wave one (wave pairs, two per SIMD).  The two are
    // bit-identical.  Before the epilogue's residual loads were batched they were equally fast (1.39 vs 1.38 ms at M = 460 800, stage-1 line 2.2823 vs 2.2824
    // frames/s); with the batched epilogue the eight-wave form is ahead (1.24 vs 1.31 ms; same-box stage-1 line 2.336 vs 2.318, AR chunk 7.48 vs 7.53 s:
    // profiles/r05_ff_fused_probe_epilogue_mlp.txt, r05_bench6_ff_waves_ab.txt) and is the default.
    static const bool four_waves = [] { const char* e = getenv("SVD_FF_WAVES"); return e && e[0] == '4'; }();
#define FF_LAUNCH(RES, OUT)  FF_LAUNCH3(RES, OUT, false, false)
#define FF_LAUNCH2(RES, OUT, BL)  FF_LAUNCH3(RES, OUT, BL, false)
#define FF_LAUNCH3(RES, OUT, BL, LNV)                                                                                                       \
    do {                                                                                                                                 \
        SVD_DISPATCH_DTYPE(dtype, {                                                                                                      \
            if (four_waves FF_FORCE4) {          /* probe builds: the variant alone decides */                                                                                                  \
                auto kern = ff_geglu_fused_kernel<E, RES, OUT, BL, 0, LNV>;                                                                       \
                static unsigned char attr_set_dev[64] = {0};                                                                             \
                unsigned char& attr_set = attr_set_dev[dev];                                                                             \
                if (!attr_set) {                                                                                                         \
                    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
                    attr_set = true;                                                                                                     \
                }                                                                                                                        \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), FF_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, \
                                   alpha, Y, ldy, (int)M, ntiles, rowvec, rowvec_ld, rows_per_vec, ln_gamma, ln_beta, ln_eps, ln_addvec, ln_addvec_ld, ln_rows_per_vec, Yn, ldyn, stagger_ticks, stagger_from);                                                                                      \
            } else {                                                                                                                     \
                auto kern = ff_geglu_fused8_kernel<E, RES, OUT, BL, 0, LNV>;                                                                      \
                static unsigned char attr_set_dev[64] = {0};                                                                             \
                unsigned char& attr_set = attr_set_dev[dev];                                                                             \
                if (!attr_set) {                                                                                                         \
                    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
                    attr_set = true;                                                                                                     \
                }                                                                                                                        \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), F8_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, \
                                   alpha, Y, ldy, (int)M, ntiles, rowvec, rowvec_ld, rows_per_vec, ln_gamma, ln_beta, ln_eps, ln_addvec, ln_addvec_ld, ln_rows_per_vec, Yn, ldyn, stagger_ticks, stagger_from);                                                                                      \
            }                                                                                                                            \
        });                                                                                                                              \
    } while (0)
#ifdef SVD_FF_PROBES
    if (svd_ff_probe_variant > 0) {
#define FF_PROBE(PVV)                                                                                                                    \
        do {                                                                                                                             \
            auto kern = ff_geglu_fused_kernel<ElemF16, 2, true, false, PVV>;                                                                   \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), FF_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, alpha, Y, ldy, (int)M, ntiles, rowvec, rowvec_ld, rows_per_vec, ln_gamma, ln_beta, ln_eps, ln_addvec, ln_addvec_ld, ln_rows_per_vec, Yn, ldyn, stagger_ticks, stagger_from); \
        } while (0)
#define FF_PROBE8(PVV)                                                                                                                   \
        do {                                                                                                                             \
            auto kern = ff_geglu_fused8_kernel<ElemF16, 2, true, false, PVV>;                                                              \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, F8_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), F8_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, nch, b2, R, ldr, S, lds, alpha, Y, ldy, (int)M, ntiles, rowvec, rowvec_ld, rows_per_vec, ln_gamma, ln_beta, ln_eps, ln_addvec, ln_addvec_ld, ln_rows_per_vec, Yn, ldyn, stagger_ticks, stagger_from); \
        } while (0)
        switch (svd_ff_probe_variant) {
            case 1: FF_PROBE(1); break; case 2: FF_PROBE(2); break; case 3: FF_PROBE(3); break; case 4: FF_PROBE(4); break;
            case 101: FF_PROBE8(1); break; case 102: FF_PROBE8(2); break; case 104: FF_PROBE8(4); break; case 108: FF_PROBE8(8); break;
            case 116: FF_PROBE8(16); break; case 115: FF_PROBE8(15); break; case 107: FF_PROBE8(7); break; case 103: FF_PROBE8(3); break;
            default: return SVD_EINVAL;
        }
        SVD_CHECK_LAUNCH("ff_geglu_fused(probe)");
        return SVD_OK;
    }
#endif
    const int res = R ? (res_f32 ? 2 : 1) : 0;
    if (Yn) {
        FF_LAUNCH3(2, true, false, true);
    } else if (S) {
        if (res == 2) { if (out_f32) FF_LAUNCH2(2, true, true); else FF_LAUNCH2(2, false, true); }
        else { if (out_f32) FF_LAUNCH2(1, true, true); else FF_LAUNCH2(1, false, true); }
    } else if (res == 2) { if (out_f32) FF_LAUNCH(2, true); else FF_LAUNCH(2, false); }
    else if (res == 1) { if (out_f32) FF_LAUNCH(1, true); else FF_LAUNCH(1, false); }
    else { if (out_f32) FF_LAUNCH(0, true); else FF_LAUNCH(0, false); }
#undef FF_LAUNCH
#undef FF_LAUNCH2
#undef FF_LAUNCH3
    SVD_CHECK_LAUNCH("ff_geglu_fused");
    return SVD_OK;
}
