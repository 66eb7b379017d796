// Row-resident 320 -> N projection (round 6):   Y[M, N] = X[M, 320] . W^T + bias     (16-bit rows in, 16-bit rows out)
//
// Replaces svd_gemm for the q | k and q | k | v projections of the 320-channel transformer blocks -- attn1.to_q / to_k of SpatialVideoTransformer (N = 640) and
// to_q / to_k / to_v of its time_stack (N = 960): code/models/svd/sgm/modules/attention.py:246-262, video_attention.py:125-168; the enhancer's Transformer2DModel /
// TransformerTemporalModel likewise -- whose rows are the widest of the network (M = 460 800 in stage 1, 1 094 400 in the enhancer) and whose K is the shortest.
//
// Why a separate kernel.  These GEMMs are HBM-side (per token row 640 B in, 1 280 / 1 920 B out, 0.4 / 0.6 MFLOP), and the 256 x 320 tile of gemm_impl.inc reads
// its A rows once PER N TILE: the two or three workgroups that share an A tile would have to run within a k-step or two of each other to meet in the 4-MiB L2, and
// nothing synchronises them -- counted 2.37 GB per launch at N = 960 against 1.18 GB algorithmic (profiles/r06_traffic_signatures.json), 474 us where the bytes
// alone need 250.  Here, as in ff_fused.hip's first GEMM, a WAVE keeps its 32 token rows in registers (80 VGPRs: the A operands of all 20 k-steps) for the whole
// width: X is read ONCE, W streams through LDS in chunks of 64 output channels (40 KiB, a host-packed fragment image copied by LDS-DMA into a ring of three slots:
// the copy of chunk c + 2 runs under the MFMAs of chunks c and c + 1), one workgroup barrier per chunk.
//   * D[token, channel] = X . W^T untransposed (X is the A operand): a lane is a channel.  The packed image interleaves the chunk's channels over its two
//     32-column MFMA tiles (tile t, column c <-> channel 2 c + t), so a lane holds two ADJACENT channels of a token and one dword store per accumulator register
//     moves 2 token rows x 128 contiguous bytes (the access pattern tools/access_pattern_bench.hip measures fastest).
//   * 8 waves per workgroup (256 rows, two waves per SIMD), 120 KiB of LDS; the stores of two chunks stay in flight across the barriers (in-order vmcnt, counted).
//     (First form: 4 waves, two slots, two workgroups per CU: 3.0 TB/s -- a chunk's copy, requested one step ahead, had not landed when the step began.)
// Work per 256-row tile: N / 64 chunks x 40 MFMAs per wave against 160 KiB in + 256 N / 320 KiB out: HBM-bound at either width.
#include "svd_common.h"
#include <stdlib.h>

namespace {

constexpr int RP_K = 320;
constexpr int RP_NS = RP_K / 16;                    // 20 k-steps
constexpr int RP_CH = 64;                           // output channels per chunk (two MFMA column tiles)
constexpr int RP_CHUNK_BYTES = RP_NS * 2 * 1024;    // 40 fragments of 1 KiB: fragment 2 s + t
constexpr int RP_DEFAULT_RB = 1;          // measured faster than 2 (see the kernel); SVD_ROWPROJ_RB selects
constexpr int RP_SLOTS = 3;                         // LDS ring: the copy of chunk g + 2 runs under the MFMAs of chunks g and g + 1
constexpr int RP_LDS_TOTAL = RP_SLOTS * RP_CHUNK_BYTES;    // 120 KiB: one workgroup per CU
constexpr int RP_ROWS = 256;                        // token rows per tile, either form

// Fragment (k-step s, tile t) of chunk ch, lane l: W[64 ch + 2 (l % 32) + t][16 s + 8 (l / 32) .. + 7]  (video_model.pack_rowproj320)
//
// RB = 32-row blocks per wave.  RB = 1 (default): eight waves (two per SIMD), each fragment read from LDS feeds one MFMA.  RB = 2 (SVD_ROWPROJ_RB=2): four waves (one
// per SIMD) own 64 rows each, every fragment feeds TWO MFMAs -- half the LDS reads per flop, accumulators in AGPRs, the k-loop is back-to-back MFMAs.  Measured
// (profiles/r06_rowproj_probe.txt): RB = 2 is 5-8 % SLOWER (316 | 426 us against 298 | 395 us at M = 460 800, N = 640 | 960): the LDS port was not the limit, and one
// wave per SIMD has nothing to run while it waits at the chunk barrier or packs its 64 accumulators.  Both forms sit at ~0.7 PFLOP/s -- about half of what the chip's
// best GEMM sustains at its power-limited clock (MI355X_MICROARCH.md: 1.25-1.35 PFLOP/s) -- not at the 4.5 TB/s the bytes would allow.
template <class E, int RB>
__global__ __launch_bounds__(64 * (8 / RB), 1) void rowproj320_kernel(const svd_bf16* __restrict__ X, int64_t ldx, const char* __restrict__ Wp,
                                                                      const float* __restrict__ bias, svd_bf16* __restrict__ Y, int64_t ldy, int M, int ntiles, int nch) {
    constexpr int WAVES = 8 / RB, PIECES = 40 / WAVES, STORES = 16 * RB;
    constexpr int INFLIGHT = PIECES + 2 * STORES < 63 ? PIECES + 2 * STORES : 63;       // vmcnt is a 6-bit counter
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t sbase = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const uint32_t voff = (uint32_t)lane * 16u;

    // LDS-DMA of one chunk: piece p (1 KiB) by wave p % WAVES (asm volatile without a memory clobber, like ff_fused.hip: the target slot is fenced off from its
    // readers by the workgroup barriers either side)
    auto dma_chunk = [&](int ch, int slot) __attribute__((always_inline)) {
        const char* src = Wp + (int64_t)ch * RP_CHUNK_BYTES;
        const uint32_t dst = sbase + slot * RP_CHUNK_BYTES;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = wave + WAVES * i;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src + p * 1024), "s"(dst + p * 1024) : "m0");
        }
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int total = ((ntiles - 1 - tile) / (int)gridDim.x + 1) * nch;       // chunk steps of this workgroup; step g reads slot g % 3, chunk g % nch
    int g = 0, slot = 0;
    dma_chunk(0, 0);
    if (total > 1) dma_chunk(1 % nch, 1);
    for (; tile < ntiles; tile += gridDim.x) {
        // ---- the wave's rows: lane (row l31, half hi) of block rb holds channels 16 s + 8 hi .. + 7 of k-step s; rows past M re-read row M - 1 (their stores are duplicates)
        uint4 xf[RB][RP_NS];
        int64_t orow[RB][16];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r0 = tile * RP_ROWS + (wave * RB + rb) * 32;
            int row = r0 + l31;
            row = row < M ? row : M - 1;
            const svd_bf16* xp = X + (int64_t)row * ldx + 8 * hi;
#pragma unroll
            for (int s = 0; s < RP_NS; ++s) xf[rb][s] = *(const uint4*)(xp + 16 * s);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = r0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                orow[rb][r] = (int64_t)(t < M ? t : M - 1) * ldy;
            }
        }
        for (int c = 0; c < nch; ++c, ++g) {
            // vmcnt counts in order.  A step issues [PIECES pieces of chunk g + 2] [STORES stores of chunk g]; chunk g's pieces were issued two steps ago, so "all but
            // the PIECES + 2 STORES youngest operations are done" = they have landed while the stores of the last two chunks stay in flight.  The first chunk of a
            // tile also needs the tile's rows (the youngest operations), and the last two steps of the workgroup issue no pieces: everything.
            if (c == 0 || g + 2 >= total) svd_wait_dma();
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(INFLIGHT) : "memory");
            __syncthreads();                         // everyone's pieces of chunk g have landed; and every wave has finished reading the slot of chunk g - 1
            if (g + 2 < total) {
                int c2 = c + 2; c2 = c2 >= nch ? c2 - nch : c2; c2 = c2 >= nch ? c2 - nch : c2;       // the next tile starts with chunk 0 again (nch may be 1)
                int s2 = slot + 2; s2 = s2 >= RP_SLOTS ? s2 - RP_SLOTS : s2;
                dma_chunk(c2, s2);                   // into the slot of chunk g - 1
            }
            float2 b = make_float2(0.f, 0.f);
            if (bias) b = *(const float2*)(bias + c * RP_CH + 2 * l31);
            f32x16_t a[RB][2];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int i = 0; i < 16; ++i) { a[rb][0][i] = 0.f; a[rb][1][i] = 0.f; }
            const char* wl = smem + slot * RP_CHUNK_BYTES + lane * 16;
            // fragments three k-steps ahead through a ring of eight registers: an LDS read under load takes longer than the MFMAs of a k-step
            constexpr int AH = 3;
            uint4 fr[8];
#pragma unroll
            for (int i = 0; i < 2 * AH; ++i) fr[i] = *(const uint4*)(wl + i * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < RP_NS; ++s) {
                if (s + AH < RP_NS) {
                    fr[(2 * (s + AH)) & 7] = *(const uint4*)(wl + (2 * (s + AH)) * 1024);
                    fr[(2 * (s + AH) + 1) & 7] = *(const uint4*)(wl + (2 * (s + AH) + 1) * 1024);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    a[rb][0] = E::mfma(xf[rb][s], fr[(2 * s) & 7], a[rb][0]);
                    a[rb][1] = E::mfma(xf[rb][s], fr[(2 * s + 1) & 7], a[rb][1]);
                }
                __builtin_amdgcn_sched_barrier(0);          // pins the read-ahead distance (the scheduler otherwise sinks every read to its use)
            }
            uint32_t* yp = (uint32_t*)(Y + c * RP_CH) + l31;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) *(uint32_t*)((svd_bf16*)yp + orow[rb][r]) = E::pack(a[rb][0][r] + b.x, a[rb][1][r] + b.y);
            slot = slot + 1 == RP_SLOTS ? 0 : slot + 1;
        }
    }
    svd_wait_dma();
}

}  // namespace

extern "C" int64_t svd_rowproj320_pack_bytes(int32_t N) { return (N > 0 && N % RP_CH == 0) ? (int64_t)(N / RP_CH) * RP_CHUNK_BYTES : (int64_t)SVD_EINVAL; }

extern "C" int svd_rowproj320(const svd_bf16* X, int64_t ldx, const void* Wp, const float* bias, svd_bf16* Y, int64_t ldy, int64_t M, int32_t N, int32_t dtype,
                              svd_stream_t stream) {
    if (!X || !Wp || !Y || M <= 0 || M > 0x7fffff00 || N <= 0 || N % RP_CH || N > 8192) return SVD_EINVAL;
    if (ldx % 8 || ldx < RP_K || ldy % 2 || ldy < N || (((uintptr_t)X | (uintptr_t)Wp) & 15) || ((uintptr_t)Y & 3) || (bias && ((uintptr_t)bias & 7))) return SVD_EINVAL;
    const int ntiles = (int)((M + RP_ROWS - 1) / RP_ROWS);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SVD_ELAUNCH;
#define RP_LAUNCH(RBV)                                                                                                                    \
    SVD_DISPATCH_DTYPE(dtype, {                                                                                                           \
        auto kern = rowproj320_kernel<E, RBV>;                                                                                            \
        static int slots_e[64] = {0};        /* resident workgroups of this instantiation, per device (the attribute belongs to the device's copy) */ \
        int slots = __atomic_load_n(&slots_e[dev], __ATOMIC_RELAXED);                                                                     \
        if (!slots) {                                                                                                                     \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS_TOTAL) != hipSuccess) return SVD_ELAUNCH; \
            int cus = 0;                                                                                                                  \
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;              \
            slots = cus;                                                                                                                  \
            __atomic_store_n(&slots_e[dev], slots, __ATOMIC_RELAXED);                                                                     \
        }                                                                                                                                 \
        const int grid = ntiles < slots ? ntiles : slots;                                                                                 \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (8 / RBV)), RP_LDS_TOTAL, (hipStream_t)stream, X, ldx, (const char*)Wp, bias, Y, ldy, (int)M, ntiles, N / RP_CH); \
    })
    // SVD_ROWPROJ_RB=1|2 (A/B): 32-row blocks per wave, see the kernel
    static const int rb = [] { const char* e = getenv("SVD_ROWPROJ_RB"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : RP_DEFAULT_RB; }();
    if (rb == 1) { RP_LAUNCH(1); } else { RP_LAUNCH(2); }
#undef RP_LAUNCH
    SVD_CHECK_LAUNCH("rowproj320");
    return SVD_OK;
}
