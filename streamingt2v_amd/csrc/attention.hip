// Attention kernels for gfx950 (head dim 64 everywhere in the StreamingSVD UNet / ControlNet / CAM).
#include "svd_common.h"
#include <stdlib.h>

namespace {


// ------------------------------------------------------------------------------------------------------------
// Spatial self-attention, flash style.  One workgroup = NW waves x 32 query rows; KV tiles of 64 keys.
//   S^T[key][query] = K . Q^T        (A = K rows from LDS, B = Q rows held in registers)
//   -> a lane owns ONE query column: online softmax is lane-local (+1 cross-half exchange for the row max).
//   O^T[d][query]  += V^T . P^T      (A = V^T rows (keys contiguous) from LDS, B = P from the S^T registers)
//   -> P never leaves registers: the keys of a 32-key block are permuted (bits 2<->3 of the LDS row index) while
//      staging K so that accumulator registers 8s..8s+7 of a lane are 8 CONSECUTIVE keys, i.e. exactly the
//      B-operand fragment of the PV MFMA, matched by one ds_read_b128 of V^T.
// K and V^T tiles are staged with global_load_lds; swizzle (slot ^= (row>>1)&7) on source address and on read.
// ------------------------------------------------------------------------------------------------------------
// TPB = KV tiles per LDS buffer and per workgroup barrier.  TPB = 2 (round-3 experiment, the round-2 review's "two KV tiles per barrier"): a
// buffer holds two tiles (2 x 16 KB), the barrier and the DMA drain come every second tile and the loads run two tiles ahead; 8-wave
// workgroups only (2 x 64 KB per CU).  Measured slightly SLOWER than TPB = 1 (see the dispatcher): kept as an A/B build, not the default.
template <int NW, class E, int TPB>
__global__ __launch_bounds__(NW * 64, 4) void attn_spatial_d64_kernel(
    const svd_bf16* __restrict__ Q, int64_t ldq, const svd_bf16* __restrict__ K, int64_t ldk,
    const svd_bf16* __restrict__ Vt, int64_t tok_ld, svd_bf16* __restrict__ O, int64_t ldo,
    int frames, int n_q, int n_tok /* keys */, int kv_div, int heads, int qblocks) {
    constexpr int NT = NW * 64;
    constexpr int BQ = NW * 32;
    constexpr int KT = 64;                 // keys per tile
    constexpr int TILE_B = KT * 128;       // 8 KiB : [64 rows][64 bf16]
    constexpr int RPP = NT / 8;            // tile rows staged per pass
    constexpr int PASSES = 64 / RPP;       // 2 (NW=4) or 1 (NW=8)
    constexpr int BUF_B = TPB * 2 * TILE_B;   // one buffer: TPB x (K tile + Vt tile)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 buffers
    const uint32_t smem_base = lds_addr_of(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware block order: all query blocks of one (frame, head) stay on one XCD (K/V reuse in its L2).
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int fh = wgid / qblocks, qb = wgid - fh * qblocks;
    const int f = fh / heads, h = fh - f * heads;

    // cross-attention: n_q queries per frame attend to the n_tok keys of key/value set f / kv_div (self-attention: n_q ==
    // n_tok, kv_div == 1)
    const int fkv = f / kv_div;
    const svd_bf16* Qf = Q + (int64_t)f * n_q * ldq + h * 64;
    const svd_bf16* Kf = K + (int64_t)fkv * n_tok * ldk + h * 64;
    const svd_bf16* Vf = Vt + ((int64_t)fkv * heads + h) * 64 * tok_ld;

    // Q fragments (B operand): lane -> query l31, d = 16*ks + 8*hi .. +8
    int qrow = qb * BQ + wave * 32 + l31;
    const bool q_valid = qrow < n_q;
    if (!q_valid) qrow = n_q - 1;
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(Qf + (int64_t)qrow * ldq + 16 * ks + 8 * hi);

    // staging coordinates
    const int srow = tid >> 3, ps = tid & 7;
    const int ls = ps ^ ((srow >> 1) & 7);
    auto stage = [&](int kv0, int buf, int sub) {
        const uint32_t dK = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF_B + sub * (2 * TILE_B) + wave * 1024);
        const uint32_t dV = dK + TILE_B;
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            const int r = j * RPP + srow;                                   // LDS row
            const int kperm = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);  // key held by this row (bits 2<->3)
            int key = kv0 + kperm;
            if (key > n_tok - 1) key = n_tok - 1;
            glds16_asm(Kf + (int64_t)key * ldk + ls * 8, dK + j * (NT * 16));
            glds16_asm(Vf + (int64_t)r * tok_ld + kv0 + ls * 8, dV + j * (NT * 16));
        }
    };

    f32x16_t o_acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o_acc[0][i] = 0.f; o_acc[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float c = 0.125f * 1.44269504088896341f;   // d^-0.5 * log2(e)

    const int ntiles = (n_tok + KT - 1) / KT;
    const int ngroups = (ntiles + TPB - 1) / TPB;
    auto stage_group = [&](int g, int buf) {
#pragma unroll
        for (int u = 0; u < TPB; ++u)
            if (g * TPB + u < ntiles) stage((g * TPB + u) * KT, buf, u);
    };
    stage_group(0, 0);
    svd_wait_dma();
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
      const int cur = g & 1;
      if (g + 1 < ngroups) stage_group(g + 1, cur ^ 1);
#pragma unroll
      for (int u = 0; u < TPB; ++u) {
        const int t = g * TPB + u;
        if (t >= ntiles) break;                    // wave-uniform: the last group may hold one tile
        const char* sK = smem + cur * BUF_B + u * (2 * TILE_B);
        const char* sV = sK + TILE_B;

        // ---- S^T = K Q^T ----
        f32x16_t s_acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { s_acc[0][i] = 0.f; s_acc[1][i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int lslot = 2 * ks + hi;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int row = kb * 32 + l31;
                const uint4 kf = *(const uint4*)(sK + row * 128 + ((lslot ^ ((row >> 1) & 7)) << 4));
                s_acc[kb] = E::mfma(kf, qf[ks], s_acc[kb]);
            }
        }
        // mask keys beyond n_tok (last tile only).  register r of block kb, half hi <-> key
        //   kv0 + 32*kb + (r&7) + 8*hi + 16*(r>>3)
        const int kv0 = t * KT;
        if (kv0 + KT > n_tok) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * kb + (r & 7) + 8 * hi + 16 * (r >> 3);
                    if (key >= n_tok) s_acc[kb][r] = -INFINITY;
                }
        }
        // ---- online softmax (lane-local row) ----
        float mx = s_acc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s_acc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        // lazy rescale: once the running maximum has settled (after the first few KV tiles for typical scores) alpha is exactly
        // 1 in every lane; the wave-uniform test skips the 32 accumulator multiplies of the tile
        const bool rescale = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0;
        m_run = m_new;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): the softmax is VALU-bound (32 exp2 + ~110 other VALU ops per tile
        // against 16 MFMAs), so two scores per instruction where the ISA has it
        uint32_t pk[2][8];
#ifdef SVD_ATTN_SCALAR_SOFTMAX   /* A/B build (profiles/r02_attn_packed_vs_scalar.txt: +1 % = noise): single-issue fp32 ops instead of the packed forms */
        const float mneg = -m_new;
        float psa = 0.f, psb = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                float e0, e1;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(s_acc[kb][r]), "v"(c), "v"(mneg));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(s_acc[kb][r + 1]), "v"(c), "v"(mneg));
                const float p0 = __builtin_amdgcn_exp2f(e0), p1 = __builtin_amdgcn_exp2f(e1);
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(psa) : "v"(psa), "v"(p0));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(psb) : "v"(psb), "v"(p1));
                pk[kb][r >> 1] = E::pack(p0, p1);
            }
        l_run = l_run * alpha + (psa + psb);
#else
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 c2 = {c, c}, m2 = {-m_new, -m_new};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 sv = {s_acc[kb][r], s_acc[kb][r + 1]};
                const f32x2 e = __builtin_elementwise_fma(sv, c2, m2);
                f32x2 pv;
                pv[0] = __builtin_amdgcn_exp2f(e[0]);
                pv[1] = __builtin_amdgcn_exp2f(e[1]);
                ps2 += pv;
                pk[kb][r >> 1] = E::pack(pv[0], pv[1]);
            }
        l_run = l_run * alpha + (ps2[0] + ps2[1]);
#endif
        if (rescale) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o_acc[0][i] *= alpha; o_acc[1][i] *= alpha; }
        }

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int lslot = 2 * s + hi;
            uint4 pf;
            pf.x = pk[s >> 1][4 * (s & 1) + 0]; pf.y = pk[s >> 1][4 * (s & 1) + 1];
            pf.z = pk[s >> 1][4 * (s & 1) + 2]; pf.w = pk[s >> 1][4 * (s & 1) + 3];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row = db * 32 + l31;
                const uint4 vf = *(const uint4*)(sV + row * 128 + ((lslot ^ ((row >> 1) & 7)) << 4));
                o_acc[db] = E::mfma(vf, pf, o_acc[db]);
            }
        }
      }
      svd_wait_dma();
      __syncthreads();
    }

    // ---- finalize & store: lane holds query l31, d = 32*db + 8*g + 4*hi + (0..3) ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (q_valid) {
        svd_bf16* Orow = O + ((int64_t)f * n_q + qrow) * ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = E::pack(o_acc[db][4 * g + 0] * inv, o_acc[db][4 * g + 1] * inv);
                o.y = E::pack(o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv);
                *(uint2*)(Orow + 32 * db + 8 * g + 4 * hi) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Round-4 experiment (SVD_ATTN_PIPE=1, 8-wave workgroups only): the SAME arithmetic with the S^T MFMAs of KV tile t+1 issued BEFORE the softmax of
// tile t inside a wave, so that a wave's own matrix work runs under its own softmax VALU (the round-3 review's untried variant).  That needs a second
// score accumulator (32 registers): 2 waves per SIMD with up to 256 VGPRs instead of 4 with 128, a ring of THREE K / V^T buffers (tile t+2 is requested
// while t+1 is multiplied and t is normalised) and sched_group_barrier to interleave the 8 MFMAs with the VALU stream in program order (the wave
// issues in order: MFMAs placed first would stall the VALU behind their dependent chain).
// ------------------------------------------------------------------------------------------------------------
template <class E>
__global__ __launch_bounds__(512, 2) void attn_spatial_pipe_kernel(
    const svd_bf16* __restrict__ Q, int64_t ldq, const svd_bf16* __restrict__ K, int64_t ldk,
    const svd_bf16* __restrict__ Vt, int64_t tok_ld, svd_bf16* __restrict__ O, int64_t ldo,
    int frames, int n_q, int n_tok /* keys */, int kv_div, int heads, int qblocks) {
    constexpr int NW = 8, NT = NW * 64, BQ = NW * 32, KT = 64, TILE_B = KT * 128, BUF_B = 2 * TILE_B, NBUF = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF buffers of (K tile + Vt tile)
    const uint32_t smem_base = lds_addr_of(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int wgid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int fh = wgid / qblocks, qb = wgid - fh * qblocks;
    const int f = fh / heads, h = fh - f * heads;
    const int fkv = f / kv_div;
    const svd_bf16* Qf = Q + (int64_t)f * n_q * ldq + h * 64;
    const svd_bf16* Kf = K + (int64_t)fkv * n_tok * ldk + h * 64;
    const svd_bf16* Vf = Vt + ((int64_t)fkv * heads + h) * 64 * tok_ld;
    int qrow = qb * BQ + wave * 32 + l31;
    const bool q_valid = qrow < n_q;
    if (!q_valid) qrow = n_q - 1;
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(Qf + (int64_t)qrow * ldq + 16 * ks + 8 * hi);
    const int srow = tid >> 3, ps = tid & 7;          // NT / 8 = 64 rows per pass: one pass per tile
    const int ls = ps ^ ((srow >> 1) & 7);
    auto stage = [&](int t, int buf) {
        const int kv0 = t * KT;
        const uint32_t dK = __builtin_amdgcn_readfirstlane(smem_base + buf * BUF_B + wave * 1024);
        const int r = srow;
        const int kperm = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);
        int key = kv0 + kperm;
        if (key > n_tok - 1) key = n_tok - 1;
        glds16_asm(Kf + (int64_t)key * ldk + ls * 8, dK);
        glds16_asm(Vf + (int64_t)r * tok_ld + kv0 + ls * 8, dK + TILE_B);
    };
    auto scores = [&](int buf, f32x16_t (&s_acc)[2]) __attribute__((always_inline)) {
        const char* sK = smem + buf * BUF_B;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s_acc[0][i] = 0.f; s_acc[1][i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int lslot = 2 * ks + hi;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int row = kb * 32 + l31;
                const uint4 kf = *(const uint4*)(sK + row * 128 + ((lslot ^ ((row >> 1) & 7)) << 4));
                s_acc[kb] = E::mfma(kf, qf[ks], s_acc[kb]);
            }
        }
    };
    f32x16_t o_acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o_acc[0][i] = 0.f; o_acc[1][i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float c = 0.125f * 1.44269504088896341f;
    const int ntiles = (n_tok + KT - 1) / KT;
    stage(0, 0);
    if (ntiles > 1) stage(1, 1);
    svd_wait_dma();
    __syncthreads();
    f32x16_t s_a[2], s_b[2];
    scores(0, s_a);
    int bcur = 0;                                      // buffer of tile t
    // one KV tile: `sc` holds S^T of tile t (computed one iteration earlier), `sn` receives S^T of tile t+1.  The two accumulator sets swap roles
    // every tile (the loop below is unrolled by two) instead of being copied.
    auto tile = [&](int t, f32x16_t (&sc)[2], f32x16_t (&sn)[2]) __attribute__((always_inline)) {
        const int bnxt = bcur + 1 == NBUF ? 0 : bcur + 1, bnn = bnxt + 1 == NBUF ? 0 : bnxt + 1;
        if (t + 2 < ntiles) stage(t + 2, bnn);          // the buffer of tile t-1: every wave is past the barrier that ended iteration t-1
        const int kv0 = t * KT;
        if (kv0 + KT > n_tok) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * kb + (r & 7) + 8 * hi + 16 * (r >> 3);
                    if (key >= n_tok) sc[kb][r] = -INFINITY;
                }
        }
        // ---- S^T of tile t+1 (MFMA) interleaved with the softmax of tile t (VALU).  Unconditional: after the last tile the buffer holds stale but
        //      valid data and the result is never used -- a branch here would put the MFMAs in their own basic block, out of the scheduler's reach.
        float mx = sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const bool rescale = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0;
        m_run = m_new;
        if (rescale) {          // (before the MFMAs of this tile: the compiler hoists this branch here anyway -- it only needs the row maximum)
#pragma unroll
            for (int i = 0; i < 16; ++i) { o_acc[0][i] *= alpha; o_acc[1][i] *= alpha; }
        }
        scores(bnxt, sn);       // same basic block as the exponentials below: the scheduler interleaves these independent MFMAs with them
        uint32_t pk[2][8];
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 c2 = {c, c}, m2 = {-m_new, -m_new};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 sv = {sc[kb][r], sc[kb][r + 1]};
                const f32x2 e = __builtin_elementwise_fma(sv, c2, m2);
                f32x2 pv;
                pv[0] = __builtin_amdgcn_exp2f(e[0]);
                pv[1] = __builtin_amdgcn_exp2f(e[1]);
                ps2 += pv;
                pk[kb][r >> 1] = E::pack(pv[0], pv[1]);
            }
        l_run = l_run * alpha + (ps2[0] + ps2[1]);
        // ---- O^T += V^T P^T ----
        const char* sV = smem + bcur * BUF_B + TILE_B;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int lslot = 2 * s4 + hi;
            uint4 pf;
            pf.x = pk[s4 >> 1][4 * (s4 & 1) + 0]; pf.y = pk[s4 >> 1][4 * (s4 & 1) + 1];
            pf.z = pk[s4 >> 1][4 * (s4 & 1) + 2]; pf.w = pk[s4 >> 1][4 * (s4 & 1) + 3];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int row = db * 32 + l31;
                const uint4 vf = *(const uint4*)(sV + row * 128 + ((lslot ^ ((row >> 1) & 7)) << 4));
                o_acc[db] = E::mfma(vf, pf, o_acc[db]);
            }
        }
        bcur = bnxt;
        svd_wait_dma();
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile(t, s_a, s_b);
        if (t + 1 < ntiles) tile(t + 1, s_b, s_a);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (q_valid) {
        svd_bf16* Orow = O + ((int64_t)f * n_q + qrow) * ldo + h * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 o;
                o.x = E::pack(o_acc[db][4 * g + 0] * inv, o_acc[db][4 * g + 1] * inv);
                o.y = E::pack(o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv);
                *(uint2*)(Orow + 32 * db + 8 * g + 4 * hi) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Per-pixel temporal attention (sequence <= 32 frames), head dim 64.  HBM-bound: q/k/v are read once, in place,
// from the (frame, pixel, channel) token layout -- the reference's "(b t) s c -> (b s) t c" transposes never
// happen.  One half-wave (32 lanes) per (batch, pixel, head): lane i holds query i in fp32 registers; K and V of
// the problem sit in LDS (bf16) and are read as wave-broadcast 16-byte vectors.
// ------------------------------------------------------------------------------------------------------------
// Bound: NOT HBM.  Per (pixel, head) this VALU form does 2 * T * T * 64 fp32 FMAs and re-reads every K / V row from LDS once per query
// lane: measured 1.5-1.8 TB/s of algorithmic traffic.  Sequences of <= 32 frames (all of StreamingSVD's: 25 x 25, CAM 25 x 7) now go to
// attn_temporal_mfma_kernel below (4.5-5.0 TB/s, profiles/r02_attn_temporal_mfma_vs_valu.txt); this kernel serves the longer windows of
// the enhancer and stays selectable for T <= 32 as the A/B reference (SVD_ATTN_TEMPORAL_VALU=1).
// Three instantiations: <32 keys, 32 lanes, 8 problems per workgroup> (SVD: 25 frames, CAM 25 x 7), <64, 64, 4> (the enhancer's
// 38-frame windows) and <128, 64, 2> (the enhancer without blending: ONE window of up to 128 frames, the reference's default
// when --use_randomized_blending is not given).  LANES lanes serve one problem; queries beyond LANES are handled in passes that
// share the staged K/V.
template <class E, int TA_MAXT, int LANES, int NG>
__global__ __launch_bounds__(LANES * NG) void attn_temporal_d64_kernel(
    const svd_bf16* __restrict__ Q, int64_t ldq, const svd_bf16* __restrict__ K, int64_t ldk,
    const svd_bf16* __restrict__ V, int64_t ldv, svd_bf16* __restrict__ O, int64_t ldo,
    int batch, int tq, int tk, int n_pix, int heads, int64_t n_prob) {
    __shared__ __attribute__((aligned(16))) uint16_t sKV[NG][2][TA_MAXT][64];   // 64 KiB
    const int tid = threadIdx.x;
    const int hw = tid / LANES;         // lane group = problem slot
    const int li = tid % LANES;
    const float c = 0.125f;

    for (int64_t base = (int64_t)blockIdx.x * NG; base < n_prob; base += (int64_t)gridDim.x * NG) {
        const int64_t prob = base + hw;
        const bool active = prob < n_prob;
        // problem index -> (b, p, h) with h fastest: the lane groups of a block read adjacent channels/pixels
        int b = 0, pp = 0, h = 0;
        if (active) {
            h = (int)(prob % heads);
            const int64_t bp = prob / heads;
            pp = (int)(bp % n_pix); b = (int)(bp / n_pix);
        }
        __syncthreads();   // previous iteration's LDS reads done
        if (active) {
            // stage K,V: tk rows x 128 B each = tk*8 16-byte vectors per matrix, spread over the group's lanes
            for (int v = li; v < tk * 8; v += LANES) {
                const int j = v >> 3, sl = v & 7;
                const int64_t row = ((int64_t)b * tk + j) * n_pix + pp;
                *(uint4*)&sKV[hw][0][j][sl * 8] = *(const uint4*)(K + row * ldk + h * 64 + sl * 8);
                *(uint4*)&sKV[hw][1][j][sl * 8] = *(const uint4*)(V + row * ldv + h * 64 + sl * 8);
            }
        }
        __syncthreads();
        for (int q0 = 0; q0 < tq; q0 += LANES) {       // query passes (one unless tq > LANES)
            float q[64];
            const bool qact = active && q0 + li < tq;
            {
                const int qi = q0 + li < tq ? q0 + li : tq - 1;
                const int64_t row = ((int64_t)b * tq + qi) * n_pix + pp;
                const svd_bf16* qp = Q + row * ldq + h * 64;
#pragma unroll
                for (int d = 0; d < 64; d += 8) {
                    uint4 u = active ? *(const uint4*)(qp + d) : make_uint4(0, 0, 0, 0);
                    q[d + 0] = E::lo(u.x); q[d + 1] = E::hi(u.x);
                    q[d + 2] = E::lo(u.y); q[d + 3] = E::hi(u.y);
                    q[d + 4] = E::lo(u.z); q[d + 5] = E::hi(u.z);
                    q[d + 6] = E::lo(u.w); q[d + 7] = E::hi(u.w);
                }
            }
            // scores
            float s[TA_MAXT];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < TA_MAXT; ++j) {
                if (j < tk) {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                    for (int d = 0; d < 64; d += 8) {
                        const uint4 u = *(const uint4*)&sKV[hw][0][j][d];
                        a0 += q[d + 0] * E::lo(u.x); a1 += q[d + 1] * E::hi(u.x);
                        a2 += q[d + 2] * E::lo(u.y); a3 += q[d + 3] * E::hi(u.y);
                        a0 += q[d + 4] * E::lo(u.z); a1 += q[d + 5] * E::hi(u.z);
                        a2 += q[d + 6] * E::lo(u.w); a3 += q[d + 7] * E::hi(u.w);
                    }
                    s[j] = ((a0 + a1) + (a2 + a3)) * c;
                    mx = fmaxf(mx, s[j]);
                } else s[j] = -INFINITY;
            }
            float l = 0.f;
#pragma unroll
            for (int j = 0; j < TA_MAXT; ++j) {
                if (j < tk) { s[j] = __expf(s[j] - mx); l += s[j]; }
            }
            const float inv = 1.f / l;
            // output (reuse q registers as accumulators)
#pragma unroll
            for (int d = 0; d < 64; ++d) q[d] = 0.f;
#pragma unroll
            for (int j = 0; j < TA_MAXT; ++j) {
                if (j < tk) {
                    const float pj = s[j] * inv;
#pragma unroll
                    for (int d = 0; d < 64; d += 8) {
                        const uint4 u = *(const uint4*)&sKV[hw][1][j][d];
                        q[d + 0] += pj * E::lo(u.x); q[d + 1] += pj * E::hi(u.x);
                        q[d + 2] += pj * E::lo(u.y); q[d + 3] += pj * E::hi(u.y);
                        q[d + 4] += pj * E::lo(u.z); q[d + 5] += pj * E::hi(u.z);
                        q[d + 6] += pj * E::lo(u.w); q[d + 7] += pj * E::hi(u.w);
                    }
                }
            }
            if (qact) {
                const int64_t row = ((int64_t)b * tq + q0 + li) * n_pix + pp;
                svd_bf16* op = O + row * ldo + h * 64;
#pragma unroll
                for (int d = 0; d < 64; d += 8) {
                    uint4 u;
                    u.x = E::pack(q[d + 0], q[d + 1]); u.y = E::pack(q[d + 2], q[d + 3]);
                    u.z = E::pack(q[d + 4], q[d + 5]); u.w = E::pack(q[d + 6], q[d + 7]);
                    *(uint4*)(op + d) = u;
                }
            }
        }
    }
}

// Row softmax, fp32 scores -> bf16 probabilities.  One workgroup per row; row cached in registers (n <= 16384).
template <class E>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_,
                                                           svd_bf16* __restrict__ P, int64_t ldp, int n, float scale) {
    __shared__ float red[8];
    const int64_t r = blockIdx.x;
    const float* s = S + r * lds_;
    svd_bf16* p = P + r * ldp;
    constexpr int MAXV = 16;   // 256 threads * 4 floats * 16 = 16384
    float4 v[MAXV];
    const int nv = n >> 2;
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
            v[i] = *(const float4*)(s + idx * 4);
            mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float c = scale * 1.44269504088896341f;
    const float mc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
            v[i].x = __builtin_amdgcn_exp2f(v[i].x * c - mc); v[i].y = __builtin_amdgcn_exp2f(v[i].y * c - mc);
            v[i].z = __builtin_amdgcn_exp2f(v[i].z * c - mc); v[i].w = __builtin_amdgcn_exp2f(v[i].w * c - mc);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < nv) {
            uint2 o;
            o.x = E::pack(v[i].x * inv, v[i].y * inv); o.y = E::pack(v[i].z * inv, v[i].w * inv);
            *(uint2*)(p + idx * 4) = o;
        }
    }
}

}  // namespace

extern "C" int svd_attn_cross_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                                  const svd_bf16* Vt, int64_t tok_ld, svd_bf16* O, int64_t ldo,
                                  int32_t frames, int32_t n_q, int32_t n_k, int32_t frames_per_kv, int32_t heads,
                                  int32_t dtype, svd_stream_t stream) {
    if (!Q || !K || !Vt || !O || frames <= 0 || n_q <= 0 || n_k <= 0 || heads <= 0 || frames_per_kv <= 0) return SVD_EINVAL;
    if (frames % frames_per_kv) return SVD_EINVAL;
    if (ldq % 8 || ldk % 8 || tok_ld % 8 || ldo % 4) return SVD_EINVAL;
    if (tok_ld < ((n_k + 63) / 64) * 64) return SVD_EINVAL;   // V^T rows must cover whole 64-key tiles
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt) & 15) return SVD_EINVAL;
    // 8 waves (256 queries) per workgroup for long sequences: every workgroup streams ALL keys/values of its (frame, head) through
    // LDS, so queries per workgroup set the L2 -> LDS traffic per FLOP (4 waves: 128 FLOP/B = 7 TB/s at 900 TFLOP/s, measured to
    // be the limit); 4 waves for short sequences (more workgroups, less tail).
    const bool wide = n_q >= 2048;                     // measured: 906-927 vs 876-896 TFLOP/s at 9216 / 14400 tokens
    const int nw = wide ? 8 : 4;
    // TPB = 2 (two KV tiles per barrier) is built for A/B only (SVD_ATTN_TPB=2): measured 869-879 vs 890-898 TFLOP/s at 9216 tokens and 895-901
    // vs 907-926 at 14 400 (profiles/r03_attention_tiles_per_barrier.txt) -- with 4 waves per SIMD from two workgroups the barrier of one
    // workgroup is covered by the other, and the larger buffer only delays the first tile.  Default 1.
    static const int tpb = [] { const char* e = getenv("SVD_ATTN_TPB"); return (e && e[0] == '2') ? 2 : 1; }();
    const int qblocks = (n_q + nw * 32 - 1) / (nw * 32);
    const int64_t nwg = (int64_t)frames * heads * qblocks;
    if (nwg > 0x7fffffff) return SVD_EINVAL;
    static const bool pipe = [] { const char* e = getenv("SVD_ATTN_PIPE"); return e && e[0] == '1'; }();
    if (wide && pipe) {
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_spatial_pipe_kernel<E>), dim3((unsigned)nwg), dim3(8 * 64), 3 * 2 * 8192,
                                                     (hipStream_t)stream, Q, ldq, K, ldk, Vt, tok_ld, O, ldo, frames, n_q, n_k, frames_per_kv, heads, qblocks));
    } else if (wide && tpb == 2) {
        static bool attr_set[2] = {false, false};
        SVD_DISPATCH_DTYPE(dtype, {
            if (!attr_set[E::kId == SVD_DTYPE_F16]) {
                hipFuncSetAttribute((const void*)attn_spatial_d64_kernel<8, E, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192);
                attr_set[E::kId == SVD_DTYPE_F16] = true;
            }
            hipLaunchKernelGGL((attn_spatial_d64_kernel<8, E, 2>), dim3((unsigned)nwg), dim3(8 * 64), 8 * 8192,
                               (hipStream_t)stream, Q, ldq, K, ldk, Vt, tok_ld, O, ldo, frames, n_q, n_k, frames_per_kv, heads, qblocks);
        });
    } else if (wide)
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_spatial_d64_kernel<8, E, 1>), dim3((unsigned)nwg), dim3(8 * 64), 4 * 8192,
                                                     (hipStream_t)stream, Q, ldq, K, ldk, Vt, tok_ld, O, ldo, frames, n_q, n_k,
                                                     frames_per_kv, heads, qblocks));
    else
        SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_spatial_d64_kernel<4, E, 1>), dim3((unsigned)nwg), dim3(4 * 64), 4 * 8192,
                                                     (hipStream_t)stream, Q, ldq, K, ldk, Vt, tok_ld, O, ldo, frames, n_q, n_k,
                                                     frames_per_kv, heads, qblocks));
    SVD_CHECK_LAUNCH("attn_spatial_d64");
    return SVD_OK;
}

extern "C" int svd_attn_spatial_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                                    const svd_bf16* Vt, int64_t tok_ld, svd_bf16* O, int64_t ldo,
                                    int32_t frames, int32_t n_tok, int32_t heads, int32_t dtype, svd_stream_t stream) {
    return svd_attn_cross_d64(Q, ldq, K, ldk, Vt, tok_ld, O, ldo, frames, n_tok, n_tok, 1, heads, dtype, stream);
}

// ------------------------------------------------------------------------------------------------------------
// Per-pixel temporal attention on the matrix cores, sequences <= 64 frames (SVD: 25 x 25; CAM: 25 queries x <= 32 keys; the enhancer's
// 38-frame windows as two 32-key blocks), head dim 64.
// One wave per (batch, pixel, head) problem, no workgroup-level synchronisation:
//   S^T = K Q^T   4 MFMA 32x32x16: A = K rows, B = Q rows, both read straight from the token layout in fragment layout (lane = row
//                 l31, 16 B at d = 16 ks + 8 hi) -- no LDS; the lane ends up with ITS query's 16 scores (keys (r%4) + 4 hi + 8 (r/4)),
//                 the other 16 sit in lane + 32: the softmax is lane-local + one cross-half exchange
//   O^T = V^T P^T 4 MFMA: B = P^T is the lane's own registers 8 s .. 8 s + 7 packed to 16 bit (no data movement); A = V^T needs 8 KEYS
//                 per lane for one d, the strided direction of V: V rows go through a per-wave LDS tile (coalesced 16-byte loads in,
//                 2-byte gathers out, in exactly the key order the P registers have)
//   O rows leave through the same LDS tile so that the global stores are whole 128-byte rows.
// The VALU form above costs 2 T^2 64 fp32 FMAs + T broadcast LDS reads of K and V per query lane (3.4 % of the stage-1 job, 3x its
// HBM time); this one is 8 MFMA + ~150 VALU per problem and leaves the kernel to the memory system: q, k, v read once, o written once.
// Measured on MI355X (CFG 2 x 25 frames): level 0 (9216 pixels x 5 heads) 666 -> 242 us = 4.9 TB/s; level 1 333 -> 125 us; level 2 169 -> 59 us;
// CAM 25 x 7 keys 288 -> 162 us.
// The next problem's 12 KB are requested before the current one is computed (registers), 8 waves per CU.
// ------------------------------------------------------------------------------------------------------------
namespace {
// NKB = 32-key blocks (sequences <= 32 NKB frames; queries run in blocks of 32 against the same K / V).  NKB = 1 also keeps the NEXT problem's
// 12 KB in flight in registers; NKB = 2 (the enhancer's 38-frame windows) has no registers left for that and relies on 8 waves per CU.
template <class E, int NW, int NKB>
__global__ __launch_bounds__(NW * 64) void attn_temporal_mfma_kernel(
    const svd_bf16* __restrict__ Q, int64_t ldq, const svd_bf16* __restrict__ K, int64_t ldk,
    const svd_bf16* __restrict__ V, int64_t ldv, svd_bf16* __restrict__ O, int64_t ldo,
    int batch, int tq, int tk, int n_pix, int heads, int64_t n_prob) {
    constexpr int VROW = 144;      // bytes per staged row: 128 + 16, so that the two half-waves of a 2-byte gather (rows 4 apart) use different banks
    constexpr bool PREFETCH = NKB == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int WAVE_ROWS = NKB == 1 ? 32 : 32 * NKB + 32;      // V tile (+ 32 staging rows for O when the V tile must outlive a query block)
    char* const sV = smem + wave * (WAVE_ROWS * VROW);
    const int64_t qstep = (int64_t)n_pix * ldq, kstep = (int64_t)n_pix * ldk, vstep = (int64_t)n_pix * ldv, ostep = (int64_t)n_pix * ldo;
    const int nrow = lane >> 3, nchk = lane & 7;          // natural (row-major) view of a [32][64] tile: row nrow + 8 i, 16-byte chunk nchk
    const int nqb = (tq + 31) >> 5;                       // query blocks of this problem size (<= NKB)
    struct Tile { uint4 q[NKB][4], k[NKB][4], v[4 * NKB]; };
    struct Where { int b, px, h; };                       // 32-bit problem arithmetic (the launcher guarantees n_prob < 2^31): 64-bit scalar divisions cost ~100 SALU each
    auto where = [&](int prob) __attribute__((always_inline)) -> Where {
        const int bp = prob / heads;
        Where w; w.h = prob - bp * heads; w.b = bp / n_pix; w.px = bp - w.b * n_pix;
        return w;
    };
    // fragment rows are clamped to the last real frame (masked / discarded later)
    auto load = [&](const Where& w, Tile& t) __attribute__((always_inline)) {
        const svd_bf16* q0 = Q + ((int64_t)w.b * tq * n_pix + w.px) * ldq + w.h * 64 + 8 * hi;
        const svd_bf16* k0 = K + ((int64_t)w.b * tk * n_pix + w.px) * ldk + w.h * 64 + 8 * hi;
        const svd_bf16* v0 = V + ((int64_t)w.b * tk * n_pix + w.px) * ldv + w.h * 64 + nchk * 8;
#pragma unroll
        for (int blk = 0; blk < NKB; ++blk) {
            int rq = 32 * blk + l31, rk = 32 * blk + l31;
            if (rq > tq - 1) rq = tq - 1;
            if (rk > tk - 1) rk = tk - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                t.q[blk][ks] = *(const uint4*)(q0 + (int64_t)rq * qstep + 16 * ks);
                t.k[blk][ks] = *(const uint4*)(k0 + (int64_t)rk * kstep + 16 * ks);
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * NKB; ++i) {
            int r = nrow + 8 * i;
            if (r > tk - 1) r = tk - 1;                   // finite filler: those keys get probability exactly 0
            t.v[i] = *(const uint4*)(v0 + (int64_t)r * vstep);
        }
    };
    const float c = 0.125f * 1.44269504088896341f;        // d^-0.5 * log2(e)
    const int klim = tk - 4 * hi;                         // key of register r of block kb is 32 kb + kr(r) + 4 hi, kr(r) = (r & 3) + 8 (r >> 2): real iff 32 kb + kr(r) < klim

    Tile cur;
    const int nprob = (int)n_prob, stride = (int)gridDim.x * NW;
    int prob = (int)blockIdx.x * NW + wave;
    Where wc = where(prob < nprob ? prob : 0);
    if (PREFETCH && prob < nprob) load(wc, cur);
    for (; prob < nprob; prob += stride) {
        Tile nxt;
        const bool more = prob + stride < nprob;          // wave-uniform
        const Where wn = where(more ? prob + stride : 0);
        if constexpr (PREFETCH) { if (more) load(wn, nxt); }
        else load(wc, cur);
        // ---- V rows -> LDS (row-major)
#pragma unroll
        for (int i = 0; i < 4 * NKB; ++i) *(uint4*)(sV + (nrow + 8 * i) * VROW + nchk * 16) = cur.v[i];
        __builtin_amdgcn_wave_barrier();
        svd_bf16* const o0 = O + ((int64_t)wc.b * tq * n_pix + wc.px) * ldo + wc.h * 64 + nchk * 8;
#pragma unroll
        for (int qb = 0; qb < NKB; ++qb) {
            if (qb < nqb) {                               // wave-uniform
                // ---- S^T = K Q^T : NKB key blocks x this query block
                f32x16_t s_acc[NKB];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s_acc[kb][i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) s_acc[kb] = E::mfma(cur.k[kb][ks], cur.q[qb][ks], s_acc[kb]);
                }
                // ---- softmax of the lane's query over its 16 NKB keys + the 16 NKB of lane ^ 32
                float mx = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kr = 32 * kb + (r & 3) + 8 * (r >> 2);
                        s_acc[kb][r] = (kr < klim) ? s_acc[kb][r] : -INFINITY;
                        mx = fmaxf(mx, s_acc[kb][r]);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mneg = -mx * c;
                float psum = 0.f;
                uint32_t pk[NKB][8];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float p0 = __builtin_amdgcn_exp2f(fmaf(s_acc[kb][r], c, mneg)), p1 = __builtin_amdgcn_exp2f(fmaf(s_acc[kb][r + 1], c, mneg));
                        psum += p0 + p1;
                        pk[kb][r >> 1] = E::pack(p0, p1);
                    }
                psum += __shfl_xor(psum, 32, 64);
                const float inv = 1.f / psum;
                // ---- O^T = V^T P^T : element e of the A fragment of k-step s2 of key block kb is key 32 kb + 16 s2 + 8 (e >> 2) + (e & 3) + 4 hi --
                //      the key of P register 8 s2 + e of that block
                f32x16_t o_acc[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) { o_acc[0][i] = 0.f; o_acc[1][i] = 0.f; }
                const uint16_t* vb = (const uint16_t*)(sV + 4 * hi * VROW) + l31;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        uint4 pf;
                        pf.x = pk[kb][4 * s2 + 0]; pf.y = pk[kb][4 * s2 + 1]; pf.z = pk[kb][4 * s2 + 2]; pf.w = pk[kb][4 * s2 + 3];
#pragma unroll
                        for (int db = 0; db < 2; ++db) {
                            uint32_t w[4];
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) {
                                const int e = 2 * e2;
                                const int key0 = 32 * kb + 16 * s2 + 8 * (e >> 2) + (e & 3), key1 = key0 + 1;
                                const uint32_t lo = vb[key0 * (VROW / 2) + 32 * db], hi16 = vb[key1 * (VROW / 2) + 32 * db];
                                w[e2] = lo | (hi16 << 16);
                            }
                            uint4 vf; vf.x = w[0]; vf.y = w[1]; vf.z = w[2]; vf.w = w[3];
                            o_acc[db] = E::mfma(vf, pf, o_acc[db]);
                        }
                    }
                // ---- O rows: registers (query l31, d = 32 db + 8 g + 4 hi + 0..3) -> LDS -> whole 128-byte rows to HBM.  The staging rows follow
                //      the V tile (NKB = 1: they ARE the V tile, every lane is done with it; NKB = 2: a second region, V stays for the next query block)
                char* const sO = NKB == 1 ? sV : sV + 32 * NKB * VROW;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        uint2 o;
                        o.x = E::pack(o_acc[db][4 * g + 0] * inv, o_acc[db][4 * g + 1] * inv);
                        o.y = E::pack(o_acc[db][4 * g + 2] * inv, o_acc[db][4 * g + 3] * inv);
                        *(uint2*)(sO + l31 * VROW + (32 * db + 8 * g + 4 * hi) * 2) = o;
                    }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = nrow + 8 * i;
                    if (32 * qb + r < tq) *(uint4*)(o0 + (int64_t)(32 * qb + r) * ostep) = *(const uint4*)(sO + r * VROW + nchk * 16);
                }
                __builtin_amdgcn_wave_barrier();          // the staging rows are free again
            }
        }
        if constexpr (PREFETCH) { if (more) cur = nxt; }
        wc = wn;
    }
}
}  // namespace

extern "C" int svd_attn_temporal_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                                     const svd_bf16* V, int64_t ldv, svd_bf16* O, int64_t ldo,
                                     int32_t batch, int32_t tq, int32_t tk, int32_t n_pix, int32_t heads,
                                     int32_t dtype, svd_stream_t stream) {
    if (!Q || !K || !V || !O || batch <= 0 || n_pix <= 0 || heads <= 0) return SVD_EINVAL;
    if (tq <= 0 || tk <= 0 || tq > 128 || tk > 128) return SVD_EINVAL;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8) return SVD_EINVAL;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return SVD_EINVAL;
    const int64_t n_prob = (int64_t)batch * n_pix * heads;
    static const bool valu_only = getenv("SVD_ATTN_TEMPORAL_VALU") != nullptr;      // A/B switch: the pre-MFMA kernel for T <= 32
    if (tq <= 64 && tk <= 64 && !valu_only && n_prob < ((int64_t)1 << 31) - 256 * 8 * 4) {
        constexpr int NW = 4;
        int64_t nb = (n_prob + NW - 1) / NW;
        if (nb > 256 * 8) nb = 256 * 8;
        if (tq <= 32 && tk <= 32) {
            SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_temporal_mfma_kernel<E, NW, 1>), dim3((unsigned)nb), dim3(NW * 64), NW * 32 * 144,
                                                         (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, O, ldo, batch, tq, tk, n_pix, heads, n_prob));
        } else {      // V tile of 64 rows + 32 staging rows for O per wave
            SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_temporal_mfma_kernel<E, NW, 2>), dim3((unsigned)nb), dim3(NW * 64), NW * 96 * 144,
                                                         (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, O, ldo, batch, tq, tk, n_pix, heads, n_prob));
        }
        SVD_CHECK_LAUNCH("attn_temporal_mfma");
        return SVD_OK;
    }
    const int ng = (tq <= 32 && tk <= 32) ? 8 : (tk <= 64 ? 4 : 2);
    int64_t blocks = (n_prob + ng - 1) / ng;
    if (blocks > 256 * 16) blocks = 256 * 16;
#define SVD_TA_LAUNCH(MAXT, LANES, NG)                                                                                           \
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((attn_temporal_d64_kernel<E, MAXT, LANES, NG>), dim3((unsigned)blocks), dim3(LANES * NG), 0, \
                                                 (hipStream_t)stream, Q, ldq, K, ldk, V, ldv, O, ldo, batch, tq, tk, n_pix, heads, n_prob))
    if (ng == 8) SVD_TA_LAUNCH(32, 32, 8);
    else if (ng == 4) SVD_TA_LAUNCH(64, 64, 4);
    else SVD_TA_LAUNCH(128, 64, 2);
#undef SVD_TA_LAUNCH
    SVD_CHECK_LAUNCH("attn_temporal_d64");
    return SVD_OK;
}

extern "C" int svd_softmax_rows(const float* S, int64_t lds_, svd_bf16* P, int64_t ldp, int64_t rows, int32_t n,
                                float scale, int32_t dtype, svd_stream_t stream) {
    if (!S || !P || rows <= 0 || n <= 0 || n % 4 || n > 16384 || lds_ % 4 || ldp % 4) return SVD_EINVAL;
    if (rows > 0x7fffffff) return SVD_EINVAL;
    SVD_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL(softmax_rows_kernel<E>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S,
                                                 lds_, P, ldp, n, scale));
    SVD_CHECK_LAUNCH("softmax_rows");
    return SVD_OK;
}
