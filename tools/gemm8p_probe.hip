// Probe (NOT part of libsvdhip.so): a 256x256x64 fp16 GEMM with the 8-phase / two-wave-group schedule of the CDNA4 guide
// (/opt/skills/guides/cdna_hip_programming.md section 5 "The 256^2 8-phase template"), written on THIS repository's conventions so that it can be
// compared with the product's tiles on the same operands and, if it wins, folded into gemm_impl.inc:
//   C[M,N] (fp16) = A[M,K] . W[N,K]^T + bias[N],  fp32 accumulate, v_mfma_f32_32x32x16_f16 with the TRANSPOSED accumulator of gemm_impl.inc
//   (D[n][m]: a lane holds 4 consecutive output channels of one token row), LDS-DMA staging with the source-side XOR swizzle.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++20 -Wno-inline-asm tools/gemm8p_probe.hip -o tools/_bin/gemm8p_probe -ldl
//   run  :  tools/_bin/gemm8p_probe [path/to/libsvdhip.so]      (with the library: the product's tiles 8 / 20 / 21 / 18 on the same operands)
//
// Schedule.  8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the tile = 4 x 2 fragments of 32 x 32.  A K tile (BK = 64) is four PHASES, one
// per 64 x 32 quadrant of the wave's output, in serpentine order  p0 (m-half 0, n-half 0)  p1 (m0, n1)  p2 (m1, n1)  p3 (m1, n0):
//     read segment : ds_read the operand sub-tiles the phase needs (p0: A half 0 = 8 reads + B half 0 = 4; p1: B half 1 = 4; p2: A half 1 = 8;
//                    p3: none -- B half 0 stays in registers), issue the LDS-DMA of ONE half-tile (2 instructions per lane), s_waitcnt vmcnt(8)
//     s_barrier
//     MFMA segment : 8 MFMAs (2 x 1 fragments x 4 k-steps) at raised priority
//     s_barrier
// The two wave groups (wm = 0 | 1, one wave of each per SIMD) run ONE barrier apart: while a group's waves multiply, the other group's waves
// read LDS and issue DMA -- the matrix pipe of a SIMD always has exactly one wave feeding it, and that wave's operands are already in registers.
//
// LDS: 2 stages x (A 32 KB + B 32 KB) = 128 KB.  Rows are stored permuted so that the sub-tile a phase consumes is one contiguous 16-KB
// HALF-TILE:  A row m = wm*128 + mh*64 + r  ->  LDS row mh*128 + wm*64 + r ;  B row n = wn*64 + nh*32 + r  ->  LDS row nh*128 + wn*32 + r.
// Half-tile liveness inside K tile t:  a0 and b0 are last read in p0, b1 in p1, a1 in p2.  A half-tile is re-staged (for K tile t+2, same stage)
// no earlier than two phases after its last read (the other group reads one barrier later), one half-tile per phase:
//     (t, p2): a0(t+2)   (t, p3): b0(t+2)   (t+1, p0): b1(t+2)   (t+1, p1): a1(t+2)
// i.e. every half-tile is requested >= 5 phases (10 barrier intervals, ~2 600+ cycles) before the phase that reads it, and the load stream always
// has 4 half-tiles = 8 DMA instructions per lane in flight: ONE uniform wait, s_waitcnt vmcnt(8), in the read segment of phase q retires exactly
// what phase q+1 reads -- and it sits before a barrier that both groups pass before either reads (the guide's "one barrier more when two wave
// groups run staggered").  Beyond the last K tile the requests continue on a clamped K index (harmless re-loads into dead half-tiles), so the
// count stays uniform through the tail.
#include "../streamingt2v_amd/csrc/svd_common.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <dlfcn.h>
#include <vector>

#ifndef P8_STAGGER
#define P8_STAGGER 1      /* 0: both wave groups in lock step (A/B: what the one-barrier offset is worth) */
#endif
#ifndef P8_SETPRIO
#define P8_SETPRIO 1      /* 0: no s_setprio around the MFMA segment */
#endif
#ifndef P8_SHAPES_LONGK
#define P8_SHAPES_LONGK 0 /* 1: only the long-K shapes (variant sweeps) */
#endif

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_B = 65536, OP_B = 32768, HALF_B = 16384, PASS_B = 8192;

__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const _Float16* __restrict__ A, int64_t lda, const _Float16* __restrict__ W, int64_t ldw,
                                                        const float* __restrict__ bias, _Float16* __restrict__ C, int64_t ldc, int M, int N,
                                                        int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_base = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware tile order (N fastest inside an XCD's contiguous range)
    const int tilesN = N / BN, ntiles = (M / BM) * tilesN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (wg >= ntiles) return;
    const int m0 = (wg / tilesN) * BM, n0 = (wg % tilesN) * BN;

    // ---- staging: lane -> (row of a 64-row pass, physical 16-B slot); source = inverse row permutation + swizzled k slot ----
    const int srow = tid >> 3, ps = tid & 7;
    const _Float16* srcA[2][2];   // [half][pass]
    const _Float16* srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = h * 128 + j * 64 + srow;                 // LDS row
            const int ls = ps ^ ((rl >> 1) & 7);                    // logical k slot stored in physical slot ps
            const int ma = ((rl >> 6) & 1) * 128 + (rl >> 7) * 64 + (rl & 63);        // A: LDS row mh*128 + wm*64 + r  -> tile row wm*128 + mh*64 + r
            const int nb = ((rl >> 5) & 3) * 64 + (rl >> 7) * 32 + (rl & 31);         // B: LDS row nh*128 + wn*32 + r  -> tile row wn*64 + nh*32 + r
            srcA[h][j] = A + (int64_t)(m0 + ma) * lda + ls * 8;
            srcB[h][j] = W + (int64_t)(n0 + nb) * ldw + ls * 8;
        }
    const int nk = K / BK;
    // half-tile request: op 0 = A, 1 = B; K tile kt (clamped), into stage kt & 1
    auto request = [&](int op, int h, int kt) __attribute__((always_inline)) {
        const int kc = kt < nk ? kt : nk - 1;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + (kt & 1) * STAGE_B + op * OP_B + h * HALF_B + wave * 1024);
        const _Float16* const* src = op ? srcB[h] : srcA[h];
        glds16_asm(src[0] + kc * BK, dst);
        glds16_asm(src[1] + kc * BK, dst + PASS_B);
    };

    // ---- fragment read addresses (byte offsets inside a stage) ----
    uint32_t offA[4], offB[2], swA[4], swB[2];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        const int rl = (fm >> 1) * 128 + wm * 64 + (fm & 1) * 32 + l31;
        offA[fm] = rl * 128; swA[fm] = (rl >> 1) & 7;
    }
#pragma unroll
    for (int fn = 0; fn < 2; ++fn) {
        const int rl = fn * 128 + wn * 32 + l31;
        offB[fn] = OP_B + rl * 128; swB[fn] = (rl >> 1) & 7;
    }

    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    uint4 xa[2][4];      // A sub-tile of the current m-half: [fragment of the half][k-step]
    uint4 wb[2][4];      // B sub-tiles: [n-half][k-step]
    auto read_a = [&](int stage, int mh) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                xa[f][ks] = *(const uint4*)(smem + stage * STAGE_B + offA[mh * 2 + f] + (((2 * ks + hi) ^ swA[mh * 2 + f]) << 4));
    };
    auto read_b = [&](int stage, int nh) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wb[nh][ks] = *(const uint4*)(smem + stage * STAGE_B + offB[nh] + (((2 * ks + hi) ^ swB[nh]) << 4));
    };
    auto mfma_quadrant = [&](int mh, int nh) __attribute__((always_inline)) {
        if (P8_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                acc[nh][mh * 2 + f] = ElemF16::mfma(wb[nh][ks], xa[f][ks], acc[nh][mh * 2 + f]);
        if (P8_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };
#define SEG_BARRIER()                                   \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        __builtin_amdgcn_s_barrier();                   \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)
#define READS_DONE()                                                  \
    do {                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            \
        __builtin_amdgcn_sched_barrier(0);                            \
    } while (0)
#define DMA_WAIT8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

    // ---- prologue: K tile 0 completely, a0 / b0 of K tile 1 (the order of the steady-state request stream: a0 b0 b1 a1 per K tile) ----
    request(0, 0, 0); request(1, 0, 0); request(1, 1, 0); request(0, 1, 0);
    request(0, 0, 1); request(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything of K tile 0 has landed (this wave's share)
    SEG_BARRIER();                                          // ... and everyone's
    if (P8_STAGGER && wm == 1) SEG_BARRIER();               // group 1 runs one barrier behind group 0

    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        // ---- p0: quadrant (m0, n0); requests b1(kt + 1) ----
        read_b(st, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_a(st, 0);
        request(1, 1, kt + 1);
        DMA_WAIT8();
        SEG_BARRIER();
        READS_DONE();
        mfma_quadrant(0, 0);
        SEG_BARRIER();
        // ---- p1: quadrant (m0, n1); requests a1(kt + 1) ----
        read_b(st, 1);
        request(0, 1, kt + 1);
        DMA_WAIT8();
        SEG_BARRIER();
        READS_DONE();
        mfma_quadrant(0, 1);
        SEG_BARRIER();
        // ---- p2: quadrant (m1, n1); requests a0(kt + 2) into this K tile's stage (a0 was last read two phases ago) ----
        read_a(st, 1);
        request(0, 0, kt + 2);
        DMA_WAIT8();
        SEG_BARRIER();
        READS_DONE();
        mfma_quadrant(1, 1);
        SEG_BARRIER();
        // ---- p3: quadrant (m1, n0), B half 0 still in registers; requests b0(kt + 2) ----
        request(1, 0, kt + 2);
        DMA_WAIT8();
        SEG_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        mfma_quadrant(1, 0);
        SEG_BARRIER();
    }
    if (P8_STAGGER && wm == 0) SEG_BARRIER();               // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the clamped tail requests must not outlive the workgroup's LDS

    // ---- epilogue (probe quality: direct 8-byte stores): acc[fn][fm][4g + i] = C[m = fm*32 + l31][n = fn*32 + 8g + 4hi + i] ----
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int m = m0 + wm * 128 + fm * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + fn * 32 + 8 * g + 4 * hi;
                const float4 b = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 o;
                o.x = ElemF16::pack(acc[fn][fm][4 * g + 0] + b.x, acc[fn][fm][4 * g + 1] + b.y);
                o.y = ElemF16::pack(acc[fn][fm][4 * g + 2] + b.z, acc[fn][fm][4 * g + 3] + b.w);
                *(uint2*)(C + (int64_t)m * ldc + n) = o;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Variant K ("k-half phases"): the same two staggered wave groups, but a phase is HALF A K TILE (32 deep) of the WHOLE 128 x 64 wave tile:
// 12 ds_reads (4 A + 2 B fragments x 2 k-steps), 16 MFMAs over all 8 accumulators (dependent MFMAs 8 apart instead of 2), 2 barriers per phase
// = 4 barriers per 64 of K instead of 8.  LDS: ring of 4 sub-tile stages of 32 KB (A 256 x 32 | B 256 x 32, 64-byte rows, slot ^= (row >> 2) & 3);
// sub-tile q+3 is requested in phase q into the stage phase q-1 read -- legal because every wave retires its LDS reads (lgkmcnt(0)) BEFORE
// the barrier that ends its read segment (the guide's one-phase WAR rule) -- and s_waitcnt vmcnt(8) leaves sub-tiles q+2, q+3 in flight.
__global__ __launch_bounds__(512, 2) void gemm8k_kernel(const _Float16* __restrict__ A, int64_t lda, const _Float16* __restrict__ W, int64_t ldw,
                                                        const float* __restrict__ bias, _Float16* __restrict__ C, int64_t ldc, int M, int N,
                                                        int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t smem_base = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int tilesN = N / BN, ntiles = (M / BM) * tilesN;
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if (wg >= ntiles) return;
    const int m0 = (wg / tilesN) * BM, n0 = (wg % tilesN) * BN;
    constexpr int SUB_B = 32768, SOP_B = 16384, SPASS_B = 8192;
    const int srow = tid >> 2, ps = tid & 3;                      // a pass = 128 rows x 64 B
    const _Float16* srcA[2];
    const _Float16* srcB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rl = j * 128 + srow;
        const int ls = ps ^ ((rl >> 2) & 3);
        srcA[j] = A + (int64_t)(m0 + rl) * lda + ls * 8;
        srcB[j] = W + (int64_t)(n0 + rl) * ldw + ls * 8;
    }
    const int nsub = K / 32;
    auto request = [&](int sub) __attribute__((always_inline)) {
        const int sc = sub < nsub ? sub : nsub - 1;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + (sub & 3) * SUB_B + wave * 1024);
        glds16_asm(srcA[0] + sc * 32, dst);
        glds16_asm(srcA[1] + sc * 32, dst + SPASS_B);
        glds16_asm(srcB[0] + sc * 32, dst + SOP_B);
        glds16_asm(srcB[1] + sc * 32, dst + SOP_B + SPASS_B);
    };
    uint32_t offA[4], offB[2], swA[4], swB[2];
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) { const int rl = wm * 128 + fm * 32 + l31; offA[fm] = rl * 64; swA[fm] = (rl >> 2) & 3; }
#pragma unroll
    for (int fn = 0; fn < 2; ++fn) { const int rl = wn * 64 + fn * 32 + l31; offB[fn] = SOP_B + rl * 64; swB[fn] = (rl >> 2) & 3; }
    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    uint4 xa[4][2], wb[2][2];
    request(0); request(1); request(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    SEG_BARRIER();
    if (P8_STAGGER && wm == 1) SEG_BARRIER();
    for (int q = 0; q < nsub; ++q) {
        const char* st = smem + (q & 3) * SUB_B;
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wb[fn][ks] = *(const uint4*)(st + offB[fn] + (((2 * ks + hi) ^ swB[fn]) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xa[fm][ks] = *(const uint4*)(st + offA[fm] + (((2 * ks + hi) ^ swA[fm]) << 4));
        request(q + 3);
        DMA_WAIT8();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads retired BEFORE the barrier: the stage may be re-staged one phase later
        SEG_BARRIER();
        if (P8_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
#pragma unroll
                for (int fm = 0; fm < 4; ++fm) acc[fn][fm] = ElemF16::mfma(wb[fn][ks], xa[fm][ks], acc[fn][fm]);
        if (P8_SETPRIO) __builtin_amdgcn_s_setprio(0);
        SEG_BARRIER();
    }
    if (P8_STAGGER && wm == 0) SEG_BARRIER();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
            const int m = m0 + wm * 128 + fm * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + fn * 32 + 8 * g + 4 * hi;
                const float4 b = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                uint2 o;
                o.x = ElemF16::pack(acc[fn][fm][4 * g + 0] + b.x, acc[fn][fm][4 * g + 1] + b.y);
                o.y = ElemF16::pack(acc[fn][fm][4 * g + 2] + b.z, acc[fn][fm][4 * g + 3] + b.w);
                *(uint2*)(C + (int64_t)m * ldc + n) = o;
            }
        }
}

// plain reference for the check: one thread per output element of a sampled row set
__global__ void ref_rows_kernel(const _Float16* A, int64_t lda, const _Float16* W, int64_t ldw, const float* bias, float* out, const int* rows,
                                int nrows, int N, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * N) return;
    const int r = rows[i / N], n = i % N;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(int64_t)r * lda + k] * (float)W[(int64_t)n * ldw + k];
    out[i] = s + (bias ? bias[n] : 0.f);
}

__global__ void fill_kernel(_Float16* p, int64_t n, uint32_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (_Float16)(((int)(x & 0xffff) - 32768) / 32768.0f);        // uniform [-1, 1): full-range operands (DVFS-honest, guide rule 25)
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); exit(1); } } while (0)

float time_ms(auto&& fn, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn(); fn();
    float best = 1e9f;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

}  // namespace

int main(int argc, char** argv) {
    typedef int (*svd_gemm_fn)(const svd_gemm_args*, svd_stream_t);
    svd_gemm_fn svd_gemm_p = nullptr;
    if (argc > 1) {
        void* h = dlopen(argv[1], RTLD_NOW);
        if (!h) { printf("dlopen %s failed: %s\n", argv[1], dlerror()); return 1; }
        svd_gemm_p = (svd_gemm_fn)dlsym(h, "svd_gemm");
    }
    CK(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B));
    CK(hipFuncSetAttribute((const void*)gemm8k_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B));
#if P8_SHAPES_LONGK
    const int shapes[][3] = {{4096, 4096, 4096}, {8192, 8192, 8192}, {28672, 1280, 5120}};
    printf("variant: stagger %d setprio %d\n", P8_STAGGER, P8_SETPRIO);
#else
    const int shapes[][3] = {{512, 512, 256}, {4096, 4096, 4096}, {8192, 8192, 8192}, {28672, 1280, 5120}, {115200, 640, 2560}, {460800, 320, 1280},
                             {460800, 2560, 320}, {73728, 512, 4608}};
#endif
    int* d_rows; float* d_ref; CK(hipMalloc(&d_rows, 64 * sizeof(int)));
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        if (M % 256 || N % 256 || K % 64) { printf("skip %dx%dx%d (probe needs M, N %% 256 == 0, K %% 64 == 0)\n", M, N, K); continue; }
        _Float16 *A, *W, *C, *C2; float* bias;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2));
        CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4));
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 1u);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, W, (int64_t)N * K, 2u);
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        const int ntiles = (M / 256) * (N / 256);
        for (int variant = 0; variant < 2; ++variant) {
        auto run8p = [&] {
            if (variant == 0) hipLaunchKernelGGL(gemm8p_kernel, dim3(ntiles), dim3(512), 2 * STAGE_B, 0, A, (int64_t)K, W, (int64_t)K, bias, C, (int64_t)N, M, N, K);
            else hipLaunchKernelGGL(gemm8k_kernel, dim3(ntiles), dim3(512), 2 * STAGE_B, 0, A, (int64_t)K, W, (int64_t)K, bias, C, (int64_t)N, M, N, K);
        };
        // ---- check against the plain reference on 48 sampled rows (every output column), several runs (race screen) ----
        std::vector<int> rows;
        for (int i = 0; i < 48; ++i) rows.push_back((int)(((int64_t)i * 2654435761u + 12345) % M));
        CK(hipMemcpy(d_rows, rows.data(), 48 * sizeof(int), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_ref, (size_t)48 * N * 4));
        hipLaunchKernelGGL(ref_rows_kernel, dim3((48 * N + 255) / 256), dim3(256), 0, 0, A, (int64_t)K, W, (int64_t)K, bias, d_ref, d_rows, 48, N, K);
        std::vector<float> ref((size_t)48 * N); std::vector<_Float16> got((size_t)N);
        CK(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0.0; int bad_runs = 0;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(C, 0xff, (size_t)M * N * 2));
            run8p(); CK(hipDeviceSynchronize());
            double w = 0.0;
            for (int i = 0; i < 48; ++i) {
                CK(hipMemcpy(got.data(), C + (size_t)rows[i] * N, (size_t)N * 2, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n) {
                    const double e = fabs((double)(float)got[n] - ref[(size_t)i * N + n]) / (1.0 + fabs(ref[(size_t)i * N + n]));
                    if (!(e == e)) w = 1e9; else if (e > w) w = e;
                }
            }
            if (w > 2e-2) ++bad_runs;
            if (w > worst) worst = w;
        }
        const float ms8 = time_ms(run8p);
        printf("%7d x %5d x %5d  %s: %8.3f ms %7.1f TFLOP/s  (check: worst rel err %.2e over 4 runs, %d bad)%s", M, N, K, variant ? "k-half phases  " : "quadrant phases",
               ms8, 2.0 * M * N * K / ms8 / 1e9, worst, bad_runs, variant ? "" : "\n");
        CK(hipFree(d_ref));
        }
        if (svd_gemm_p) {
            for (int cfg : {8, 20, 21, 18}) {
                svd_gemm_args a = {};
                a.A = (const svd_bf16*)A; a.lda = K; a.W = (const svd_bf16*)W; a.ldw = K; a.M = M; a.N = N; a.K = K; a.a_mode = SVD_A_PLAIN;
                a.bias = bias; a.C = C2; a.ldc = N; a.out_mode = SVD_OUT_BF16; a.tile_cfg = cfg; a.dtype = SVD_DTYPE_F16;
                if (svd_gemm_p(&a, nullptr) != 0) { printf(" | cfg%d n/a", cfg); continue; }
                const float ms = time_ms([&] { svd_gemm_p(&a, nullptr); });
                printf(" | cfg%d %.3f ms %.0f TF", cfg, ms, 2.0 * M * N * K / ms / 1e9);
            }
        }
        printf("\n"); fflush(stdout);
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(C2)); CK(hipFree(bias));
    }
    return 0;
}
