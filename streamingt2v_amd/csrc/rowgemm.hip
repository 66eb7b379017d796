// Row-owning 320 -> 320 projection for the fp32 residual stream of the 320-channel level (round 6):
//
//     V  = R + bias + rowvec[row / rows_per_vec] + X . W^T            (fp32)           -> Y   (fp32 rows, or 16-bit rows)
//          (bias + rowvec enter as hi + lo 16-bit operands of one extra k-step: ~22 bits in fp16, 16 in bf16 -- far below the operand rounding of X . W^T)
//     Yn = LayerNorm(V) * gamma + beta                                 (eps inside the rsqrt)  -> 16-bit rows, the operand of the NEXT GEMM
//
// Replaces, for dim = 320, the svd_gemm launch of proj_in / attn1.to_out / time_stack attn1.to_out / proj_out of SpatialVideoTransformer
// (code/models/svd/sgm/modules/video_attention.py:260-333, attention.py:567-593, video_attention.py:125-168; the enhancer's Transformer2DModel /
// TransformerTemporalModel projections, code/i2v_enhance/transformer_2d.py / transformer_temporal.py) TOGETHER WITH the svd_layernorm launch that
// follows it (norm1 / norm3): the LayerNorm no longer re-reads the fp32 tensor the GEMM just wrote (1 280 B per token row of 1 920).
//
// Why a separate kernel.  These GEMMs are HBM-side: per token row 640 B of X, 1 280 B of R and 1 280 B of V against 0.2 MFLOP -- the 256 x 320 tile
// of gemm_impl.inc moves them at 2.7 TB/s (profiles/r05_roofline_table.txt: cfg 21 view 0), and its epilogue cannot normalise a row because a row's
// 320 outputs are spread over the lanes of two waves.  Here, as in ff_fused.hip, one WAVE owns 32 token rows and ALL 320 outputs of those rows:
//
//   * O^T[channel, row] = W . X^T on v_mfma_f32_32x32x16: a lane holds ITS row's outputs (160 accumulator registers, the other 160 in lane ^ 32), so
//     bias, residual, LayerNorm statistics (one cross-half exchange), affine and the 16-bit pack are lane-local;
//   * the accumulators START as the residual rows: the 40 float4 loads of R land directly in the accumulator registers (one burst of 40 KiB per wave,
//     no staging registers), the products are added on top (fp32 sum in a different order than `gemm` + residual: equal to ~1e-7 relative);
//   * the waves are INDEPENDENT: no barrier, no LDS ring, no DMA bookkeeping inside the tile loop.  Three quarters of W (the fragments of k-steps 5..19,
//     150 KiB) are copied to LDS once per workgroup and only read afterwards; the fragments of k-steps 0..4 (50 KiB) stream from L2 through a register
//     ring at the head of every tile -- all of W does not fit 160 KiB, and the ring's loads sit in the shadow of the residual burst;
//   * the four waves of a CU drift apart, so one wave's load burst overlaps the others' MFMA phases and store tails: the kernel needs no software
//     prefetch to keep ~100 KiB per CU in flight.
// Work per 32-row tile and wave: 200 MFMA (6.4 k matrix-pipe cycles) against 120 KiB of HBM traffic (19.5 us at the chip's 6.3 TB/s share): HBM-bound.
#include "svd_common.h"

namespace {

constexpr int RG_C = 320;                         // K and N
constexpr int RG_NS = RG_C / 16;                  // 20 k-steps
constexpr int RG_NO = RG_C / 32;                  // 10 output tiles of 32 channels
constexpr int RG_NF = RG_NS * RG_NO;              // 200 W fragments of 1 KiB (fragment i = s * 10 + o)
constexpr int RG_GS = 5;                          // k-steps whose fragments stream from L2 (fragments 0 .. 49)
constexpr int RG_GF = RG_GS * RG_NO;              // 50
constexpr int RG_RING = 20;                       // register ring of global fragments: two k-steps ahead
constexpr int RG_LDS_W = (RG_NF - RG_GF) * 1024;  // 153 600 B of fragments resident in LDS
constexpr int RG_LDS_G = RG_LDS_W;                // gamma | beta | bias: 3 x 1 280 B
constexpr int RG_LDS_TOTAL = RG_LDS_W + 3 * RG_C * 4;          // 157 440 B

// OUT: 0 = no Y, 1 = fp32 rows, 2 = 16-bit rows.  LN: also write Yn = LayerNorm(V).  R, rowvec: run-time (wave-uniform) options.
template <class E, int OUT, bool LN>
__global__ __launch_bounds__(256, 1) void rowgemm320_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const uint4* __restrict__ Wp,
                                                            const float* __restrict__ bias, const float* __restrict__ rowvec, int rowvec_ld,
                                                            int rows_per_vec, const float* __restrict__ R, int64_t ldr, void* __restrict__ Y,
                                                            int64_t ldy, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            svd_bf16* __restrict__ Yn, int64_t ldyn, int M, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // ---- once per workgroup: the resident part of W and the per-channel vectors -> LDS
    {
        uint4* wl = (uint4*)smem;
        const uint4* src = Wp + RG_GF * 64;
        for (int i = tid; i < (RG_NF - RG_GF) * 64; i += 256) wl[i] = src[i];
        float* vec = (float*)(smem + RG_LDS_G);
        for (int i = tid; i < RG_C; i += 256) {
            vec[i] = LN ? gamma[i] : 0.f;
            vec[RG_C + i] = LN ? beta[i] : 0.f;
            vec[2 * RG_C + i] = bias ? bias[i] : 0.f;
        }
    }
    __syncthreads();
    const char* wfrag0 = smem + lane * 16;                             // LDS fragment i (i >= 50) at wfrag + (i - 50) KiB
    const float* gam0 = (const float*)(smem + RG_LDS_G) + 4 * hi;      // this half's 4-channel groups: + 32 o + 8 j
    const uint4* wg0 = Wp + lane;                                      // global fragment i (i < 50) at wg[64 i]

    const int nw = (int)gridDim.x * 4;
    for (int tile = (int)blockIdx.x * 4 + wave; tile < ntiles; tile += nw) {
        const int row = tile * 32 + l31;
        const int rowc = row < M ? row : M - 1;                        // tail: the lanes past M recompute row M - 1 from the same inputs and store the SAME bytes to it
                                                                       // (benign duplicate stores: no exec-mask branch around each of the 60 stores)
        // W, gamma, beta and bias are the same for every tile: made opaque per iteration, or the compiler hoists their ~300 registers of loads out of
        // the tile loop and spills them (ISA audit of the first build: 80 spilled VGPRs, all of them hoisted ring / vector loads)
        // (an opaque ZERO OFFSET, not an opaque pointer: a pointer that went through an asm operand loses its address space and every LDS / global
        //  access through it becomes a flat_load, which counts on lgkmcnt AND vmcnt and forces vmcnt(0) waits)
        int zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        const char* wfrag = wfrag0 + zero;
        const float* gam = gam0 + zero;
        const uint4* wg = wg0 + zero;
        // ---- the tile's loads, one burst: the per-frame vector (consumed first: oldest in the in-order queue), residual rows straight into the
        //      accumulators, the X fragments, the first two k-steps of W
        float vec[RG_NO];
        {
            const float* rvp = rowvec ? rowvec + (int64_t)((tile * 32) / rows_per_vec) * rowvec_ld + l31 : nullptr;   // rows_per_vec % 32 == 0: one vector per tile
#pragma unroll
            for (int o = 0; o < RG_NO; ++o) vec[o] = rvp ? rvp[32 * o] : 0.f;
        }
        f32x16_t acc[RG_NO];
        if (R) {
            const float* rp = R + (int64_t)rowc * ldr + 4 * hi;
#pragma unroll
            for (int o = 0; o < RG_NO; ++o)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 r = *(const float4*)(rp + 32 * o + 8 * j);
                    acc[o][4 * j + 0] = r.x; acc[o][4 * j + 1] = r.y; acc[o][4 * j + 2] = r.z; acc[o][4 * j + 3] = r.w;
                }
        } else {
#pragma unroll
            for (int o = 0; o < RG_NO; ++o)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
        }
        uint4 xf[RG_NS];                                               // lane (row l31, half hi): channels 16 s + 8 hi .. + 7 of k-step s (B operand)
        {
            const svd_bf16* xp = X + (int64_t)rowc * ldx + 8 * hi;
#pragma unroll
            for (int s = 0; s < RG_NS; ++s) xf[s] = *(const uint4*)(xp + 16 * s);
        }
        uint4 gr[RG_RING];
#pragma unroll
        for (int i = 0; i < RG_RING; ++i) gr[i] = wg[64 * i];
        __builtin_amdgcn_sched_barrier(0);
        // bias + per-frame vector enter through the matrix pipe as well: one extra k-step whose A operand holds the vector split into two 16-bit
        // values (hi + lo: ~22 bits in fp16) and whose B operand is 1 -- 10 coalesced dword loads per tile instead of 40 strided float4 loads, and no
        // load in the epilogue at all (a load behind the epilogue's stores would wait for every one of them: vmcnt is in order)
        {
            const float* bl = gam - 4 * hi + 2 * RG_C + l31;
            uint4 ones = {0u, 0u, 0u, 0u};
            if (hi == 0) ones.x = E::pack(1.f, 1.f);
#pragma unroll
            for (int o = 0; o < RG_NO; ++o) {
                const float v = vec[o] + bl[32 * o];
                const float vh = E::lo(E::pack(v, 0.f));
                uint4 av = {0u, 0u, 0u, 0u};
                if (hi == 0) av.x = E::pack(vh, v - vh);
                acc[o] = E::mfma(av, ones, acc[o]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 200 MFMAs: k-step outer, output tile inner; fragment i = 10 s + o from the register ring (i < 50) or from LDS three reads ahead
        uint4 lr[4];
#pragma unroll
        for (int i = 0; i < RG_NF; ++i) {
            const int s = i / RG_NO, o = i % RG_NO;
            if (i + 3 >= RG_GF && i + 3 < RG_NF) lr[(i + 3) & 3] = *(const uint4*)(wfrag + (i + 3 - RG_GF) * 1024);
            if (i < RG_GF) {
                acc[o] = E::mfma(gr[i % RG_RING], xf[s], acc[o]);
                if (i + RG_RING < RG_GF) gr[i % RG_RING] = wg[64 * (i + RG_RING)];
            } else {
                acc[o] = E::mfma(lr[i & 3], xf[s], acc[o]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: lane holds row l31, channels 32 o + 8 j + 4 hi .. + 3 in acc[o][4 j .. 4 j + 3]
        float sum = 0.f;
#pragma unroll
        for (int o = 0; o < RG_NO; ++o)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = {acc[o][4 * j], acc[o][4 * j + 1], acc[o][4 * j + 2], acc[o][4 * j + 3]};
                if constexpr (LN) sum += (v.x + v.y) + (v.z + v.w);
                if constexpr (OUT == 1) {
                    *(float4*)((float*)Y + (int64_t)rowc * ldy + 32 * o + 8 * j + 4 * hi) = v;
                }
            }
        if constexpr (OUT == 2) {
            // 16-bit rows: a lane's 4 channels of group j and its partner's (lane ^ 32) are 8 consecutive channels -> one 16-byte store per PAIR of groups
            // (v_permlane32_swap: lower half keeps group j and receives the upper half's group j; upper half receives the lower half's group j + 1)
#pragma unroll
            for (int o = 0; o < RG_NO; ++o)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    uint32_t a0 = E::pack(acc[o][4 * j], acc[o][4 * j + 1]), a1 = E::pack(acc[o][4 * j + 2], acc[o][4 * j + 3]);
                    uint32_t b0 = E::pack(acc[o][4 * j + 4], acc[o][4 * j + 5]), b1 = E::pack(acc[o][4 * j + 6], acc[o][4 * j + 7]);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    uint4 w; w.x = r0[0]; w.y = r1[0]; w.z = r0[1]; w.w = r1[1];
                    *(uint4*)((svd_bf16*)Y + (int64_t)rowc * ldy + 32 * o + 8 * j + 8 * hi) = w;
                }
        }
        if constexpr (LN) {
            constexpr float invc = 1.f / (float)RG_C;
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * invc;
            float sq = 0.f;
#pragma unroll
            for (int o = 0; o < RG_NO; ++o)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float d = acc[o][i] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * invc + eps);
#pragma unroll
            for (int o = 0; o < RG_NO; ++o)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const float4 g0 = *(const float4*)(gam + 32 * o + 8 * j), g1 = *(const float4*)(gam + 32 * o + 8 * j + 8);
                    const float4 e0 = *(const float4*)(gam + RG_C + 32 * o + 8 * j), e1 = *(const float4*)(gam + RG_C + 32 * o + 8 * j + 8);
                    const float y0 = (acc[o][4 * j] - mean) * rstd * g0.x + e0.x, y1 = (acc[o][4 * j + 1] - mean) * rstd * g0.y + e0.y;
                    const float y2 = (acc[o][4 * j + 2] - mean) * rstd * g0.z + e0.z, y3 = (acc[o][4 * j + 3] - mean) * rstd * g0.w + e0.w;
                    const float y4 = (acc[o][4 * j + 4] - mean) * rstd * g1.x + e1.x, y5 = (acc[o][4 * j + 5] - mean) * rstd * g1.y + e1.y;
                    const float y6 = (acc[o][4 * j + 6] - mean) * rstd * g1.z + e1.z, y7 = (acc[o][4 * j + 7] - mean) * rstd * g1.w + e1.w;
                    const auto r0 = __builtin_amdgcn_permlane32_swap(E::pack(y0, y1), E::pack(y4, y5), false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(E::pack(y2, y3), E::pack(y6, y7), false, false);
                    uint4 w; w.x = r0[0]; w.y = r1[0]; w.z = r0[1]; w.w = r1[1];
                    *(uint4*)(Yn + (int64_t)rowc * ldyn + 32 * o + 8 * j + 8 * hi) = w;
                }
        }
    }
}

}  // namespace

// Packed image of W [320 out, 320 in]: 200 fragments of 1 KiB, fragment 10 s + o = the A operand of output tile o, k-step s: lane l holds
// W[32 o + l % 32][16 s + 8 (l / 32) .. + 7] (streamingt2v_amd/video_model.pack_rowgemm320 writes it).
extern "C" int64_t svd_rowgemm320_pack_bytes(void) { return (int64_t)RG_NF * 1024; }

extern "C" int svd_rowgemm320(const svd_bf16* X, int64_t ldx, const void* Wp, const float* bias, const float* rowvec, int32_t rowvec_ld,
                              int32_t rows_per_vec, const float* R, int64_t ldr, void* Y, int64_t ldy, int32_t out_f32, const float* ln_gamma,
                              const float* ln_beta, float ln_eps, svd_bf16* Yn, int64_t ldyn, int64_t M, int32_t dtype, svd_stream_t stream) {
    if (!X || !Wp || M <= 0 || M > 0x7fffff00 || (!Y && !Yn)) return SVD_EINVAL;
    if (ldx % 8 || ldx < RG_C || (((uintptr_t)X | (uintptr_t)Wp) & 15)) return SVD_EINVAL;
    if (R && (ldr % 4 || ldr < RG_C || ((uintptr_t)R & 15))) return SVD_EINVAL;
    if (Y && (ldy < RG_C || (out_f32 ? (ldy % 4 || ((uintptr_t)Y & 15)) : (ldy % 8 || ((uintptr_t)Y & 15))))) return SVD_EINVAL;
    if (Yn && (!ln_gamma || !ln_beta || ldyn % 8 || ldyn < RG_C || ((uintptr_t)Yn & 15))) return SVD_EINVAL;
    if (rowvec && (rows_per_vec <= 0 || rows_per_vec % 32 || rowvec_ld % 4 || ((uintptr_t)rowvec & 15))) return SVD_EINVAL;
    const int ntiles = (int)((M + 31) / 32);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SVD_ELAUNCH;
    static int n_cu_dev[64] = {0};
    if (dev < 0 || dev >= 64) return SVD_EINVAL;
    int n_cu = __atomic_load_n(&n_cu_dev[dev], __ATOMIC_RELAXED);
    if (!n_cu) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return SVD_ELAUNCH;
        n_cu = p.multiProcessorCount;
        __atomic_store_n(&n_cu_dev[dev], n_cu, __ATOMIC_RELAXED);
    }
    const int want = (ntiles + 3) / 4;
    const int grid = want < n_cu ? want : n_cu;
    const int out = !Y ? 0 : (out_f32 ? 1 : 2);
#define RG_LAUNCH(OUTM, LNM)                                                                                                              \
    do {                                                                                                                                  \
        SVD_DISPATCH_DTYPE(dtype, {                                                                                                       \
            auto kern = rowgemm320_kernel<E, OUTM, LNM>;                                                                                  \
            static unsigned char attr_set_dev[64] = {0};          /* per device: the attribute belongs to the device's copy of the function */      \
            if (!attr_set_dev[dev]) {                                                                                                      \
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
                attr_set_dev[dev] = 1;                                                                                                    \
            }                                                                                                                             \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), RG_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const uint4*)Wp, bias, rowvec, rowvec_ld,  \
                               rows_per_vec, R, ldr, Y, ldy, ln_gamma, ln_beta, ln_eps, Yn, ldyn, (int)M, ntiles);                        \
        });                                                                                                                               \
    } while (0)
    if (Yn) {
        if (out == 1) RG_LAUNCH(1, true); else if (out == 0) RG_LAUNCH(0, true); else return SVD_EINVAL;
    } else {
        if (out == 1) RG_LAUNCH(1, false); else RG_LAUNCH(2, false);
    }
#undef RG_LAUNCH
    SVD_CHECK_LAUNCH("rowgemm320");
    return SVD_OK;
}
