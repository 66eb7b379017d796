// Shared device helpers for the gfx950 kernels of libsvdhip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svdhip.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define SVD_WAVE 64

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup: ~10 VALU):
// every consumer rounds the result to 16 bit or feeds a sampler step; the GroupNorm+SiLU pass is no longer VALU-co-bound.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-erf GELU of the reference's GEGLU (attention.py:99-101):  gelu(x) = x Phi(x) = max(x, 0) - |x| q(|x|),  q(t) = Phi(-t) = 2^(r(t) - 1),
// r(t) = log2 erfc(t / sqrt 2) by a degree-6 polynomial without constant term (weighted minimax fit on [0, 7], |erf error| <= 2.3e-7; q(7) is
// 6e-13, so t is clamped there).  ONE transcendental (v_exp_f32) per element against two (rcp + exp2) for the Abramowitz-Stegun 7.1.26 form used before, no sign
// transfer, and the relative accuracy of the negative tail comes for free: |gelu error| <= 2e-6 absolute over the whole fp32 range (numpy
// float32 emulation of this exact sequence), i.e. 1/100 of the 16-bit output rounding.  The GEGLU epilogue of the ff1 GEMMs is bound by
// exactly this arithmetic (VALU is per SIMD: both waves' 64 gate values per lane serialise; round-2 ISA audit).
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float t = fminf(fabsf(x), 7.0f);
    float r = fmaf(1.775515804e-05f, t, -6.477575890e-04f);
    r = fmaf(r, t, 7.724042874e-03f);
    r = fmaf(r, t, -5.292673725e-02f);
    r = fmaf(r, t, -4.590827371e-01f);
    r = fmaf(r, t, -1.151116856e+00f);
    const float q = __builtin_amdgcn_exp2f(fmaf(r, t, -1.0f));
    return fmaf(-fabsf(x), q, fmaxf(x, 0.0f));
}

// The same function on a PAIR (round 4; the GEGLU epilogue of the ff1 GEMMs evaluates 64 of them per lane and output tile and is bound by this VALU
// work -- SQ counters: VALU pipes active 37 % of the kernel at K = 320 against 33 % MFMA-busy).  Same arithmetic, bit for bit; different instructions:
// the degree-6 polynomial runs on v_pk_fma_f32 (two elements per issue slot) instead of six v_fmaak_f32 per element, and min(|x|, 7) / max(x, 0) are
// single instructions (inline asm: the compiler brackets fminf / fmaxf with a canonicalising v_max x, x in IEEE mode -- three more VALU per element).
// Round-3 ISA: ~17 instructions per element; this form: ~10.5 (a v_pk_fma_f32 costs what two v_fma_f32 cost: the gain is the five plain ones).
typedef float svd_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ svd_f32x2 gelu_erf_f2(svd_f32x2 x) {
    svd_f32x2 t, mx, ax;
    asm("v_min_f32 %0, %2, |%1|" : "=v"(t[0]) : "v"(x[0]), "s"(7.0f));   // min(|x|, 7); the VOP3 form (|.| modifier) takes no literal: 7.0 in an SGPR
    asm("v_min_f32 %0, %2, |%1|" : "=v"(t[1]) : "v"(x[1]), "s"(7.0f));
    asm("v_max_f32 %0, 0, %1" : "=v"(mx[0]) : "v"(x[0]));                 // max(x, 0)
    asm("v_max_f32 %0, 0, %1" : "=v"(mx[1]) : "v"(x[1]));
    ax[0] = __builtin_fabsf(x[0]); ax[1] = __builtin_fabsf(x[1]);
    const svd_f32x2 c6 = {1.775515804e-05f, 1.775515804e-05f}, c5 = {-6.477575890e-04f, -6.477575890e-04f}, c4 = {7.724042874e-03f, 7.724042874e-03f},
                    c3 = {-5.292673725e-02f, -5.292673725e-02f}, c2 = {-4.590827371e-01f, -4.590827371e-01f}, c1 = {-1.151116856e+00f, -1.151116856e+00f},
                    m1 = {-1.0f, -1.0f};
    svd_f32x2 r = __builtin_elementwise_fma(c6, t, c5);
    r = __builtin_elementwise_fma(r, t, c4);
    r = __builtin_elementwise_fma(r, t, c3);
    r = __builtin_elementwise_fma(r, t, c2);
    r = __builtin_elementwise_fma(r, t, c1);
    const svd_f32x2 e = __builtin_elementwise_fma(r, t, m1);
    svd_f32x2 q;
    q[0] = __builtin_amdgcn_exp2f(e[0]); q[1] = __builtin_amdgcn_exp2f(e[1]);
    return __builtin_elementwise_fma(-ax, q, mx);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the 32 lanes of this lane's half-wave, the result in every lane.  Four of the five butterfly levels are DPP operands of the add (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: after each level the lanes a mirror pairs up hold equal partial sums, so a mirror IS the xor) -- VALU rate, no LDS crossbar, no
// lgkmcnt wait; only the exchange between the two 16-lane rows goes through ds_bpermute.  (Round 6: the LayerNorm epilogues of rowgemm320 / ff_geglu_fused reduce
// 32 values per 32-row tile this way.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_mov_f32<0xB1>(v);           // quad_perm [1,0,3,2]
    v += dpp_mov_f32<0x4E>(v);           // quad_perm [2,3,0,1]
    v += dpp_mov_f32<0x141>(v);          // row_half_mirror
    v += dpp_mov_f32<0x140>(v);          // row_mirror
    return v + __shfl_xor(v, 16, 64);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- 16-bit element types ------------------------------------------------------------------------------------
// Every kernel is templated on one of these: storage is 16 bit, arithmetic fp32.  bf16 is the default (north_star);
// fp16 is what the reference's own autocast uses (config.yaml:8 "16-mixed") and has 8x finer rounding at the same
// MFMA rate.  lo/hi: the two elements of a packed dword; pack: round-to-nearest-even.
struct ElemBF16 {
    static constexpr int kId = SVD_DTYPE_BF16;
    static __device__ __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }
    static __device__ __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        bf16x2_t t; t[0] = (__bf16)a; t[1] = (__bf16)b;            // v_cvt_pk_bf16_f32 on gfx950
        return __builtin_bit_cast(uint32_t, t);
    }
    static __device__ __forceinline__ uint16_t from_f32(float a) { return __builtin_bit_cast(uint16_t, (__bf16)a); }
    static __device__ __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
struct ElemF16 {
    static constexpr int kId = SVD_DTYPE_F16;
    static __device__ __forceinline__ float lo(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[0]; }
    static __device__ __forceinline__ float hi(uint32_t v) { return (float)__builtin_bit_cast(f16x2_t, v)[1]; }
    static __device__ __forceinline__ float to_f32(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ uint32_t pack(float a, float b) {
        f16x2_t t; t[0] = (_Float16)a; t[1] = (_Float16)b;
        return __builtin_bit_cast(uint32_t, t);
    }
    static __device__ __forceinline__ uint16_t from_f32(float a) { return __builtin_bit_cast(uint16_t, (_Float16)a); }
    static __device__ __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
#define SVD_DISPATCH_DTYPE(dtype, ...)                                     \
    do {                                                                   \
        if ((dtype) == SVD_DTYPE_BF16) { using E = ElemBF16; __VA_ARGS__; } \
        else if ((dtype) == SVD_DTYPE_F16) { using E = ElemF16; __VA_ARGS__; } \
        else return SVD_EINVAL;                                            \
    } while (0)

// LDS-DMA (global -> LDS, 16 B per lane, lane-linear destination) issued from inline asm so that hipcc does NOT
// count it: with the builtin form the compiler drains vmcnt(0) before the next ds_read of ANY LDS address, which
// serialises staging and MFMA.  The caller owns the wait: asm s_waitcnt vmcnt(0) (svd_wait_dma) + barrier before
// the staged tile is read.  `lds_dst` must be wave-uniform (byte address of the wave's 1 KiB destination).
// M0 (the LDS base of the DMA) is written and NOT restored: nothing else in these kernels reads M0 (gfx9+ LDS instructions do not need
// it; SGPR spills use immediate lanes) -- tools/check_isa.py asserts that on the built objects.  The save/restore pair cost two scalar
// issue slots per DMA instruction in the GEMM K loop.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// SADDR form: wave-uniform 64-bit base (SGPR pair) + unsigned 32-bit per-lane byte offset -- no 64-bit vector add per instruction.
__device__ __forceinline__ void glds16_saddr(uint32_t voff, const void* sbase, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void svd_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate; n > 63 waits for 63: safe, just earlier)
#define SVD_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void svd_wait_vmcnt(int n) {
    switch (n < 63 ? n : 63) {
        SVD_VMCNT_CASE(0) SVD_VMCNT_CASE(1) SVD_VMCNT_CASE(2) SVD_VMCNT_CASE(3) SVD_VMCNT_CASE(4) SVD_VMCNT_CASE(5) SVD_VMCNT_CASE(6) SVD_VMCNT_CASE(7)
        SVD_VMCNT_CASE(8) SVD_VMCNT_CASE(9) SVD_VMCNT_CASE(10) SVD_VMCNT_CASE(11) SVD_VMCNT_CASE(12) SVD_VMCNT_CASE(13) SVD_VMCNT_CASE(14) SVD_VMCNT_CASE(15)
        SVD_VMCNT_CASE(16) SVD_VMCNT_CASE(17) SVD_VMCNT_CASE(18) SVD_VMCNT_CASE(19) SVD_VMCNT_CASE(20) SVD_VMCNT_CASE(21) SVD_VMCNT_CASE(22) SVD_VMCNT_CASE(23)
        SVD_VMCNT_CASE(24) SVD_VMCNT_CASE(25) SVD_VMCNT_CASE(26) SVD_VMCNT_CASE(27) SVD_VMCNT_CASE(28) SVD_VMCNT_CASE(29) SVD_VMCNT_CASE(30) SVD_VMCNT_CASE(31)
        SVD_VMCNT_CASE(32) SVD_VMCNT_CASE(33) SVD_VMCNT_CASE(34) SVD_VMCNT_CASE(35) SVD_VMCNT_CASE(36) SVD_VMCNT_CASE(37) SVD_VMCNT_CASE(38) SVD_VMCNT_CASE(39)
        SVD_VMCNT_CASE(40) SVD_VMCNT_CASE(41) SVD_VMCNT_CASE(42) SVD_VMCNT_CASE(43) SVD_VMCNT_CASE(44) SVD_VMCNT_CASE(45) SVD_VMCNT_CASE(46) SVD_VMCNT_CASE(47)
        SVD_VMCNT_CASE(48) SVD_VMCNT_CASE(49) SVD_VMCNT_CASE(50) SVD_VMCNT_CASE(51) SVD_VMCNT_CASE(52) SVD_VMCNT_CASE(53) SVD_VMCNT_CASE(54) SVD_VMCNT_CASE(55)
        SVD_VMCNT_CASE(56) SVD_VMCNT_CASE(57) SVD_VMCNT_CASE(58) SVD_VMCNT_CASE(59) SVD_VMCNT_CASE(60) SVD_VMCNT_CASE(61) SVD_VMCNT_CASE(62)
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}
// Coarse form for hot loops: waits for vmcnt(c) with the largest c <= n from a short ladder (waiting for a few more of the OLDEST
// operations than strictly necessary is always safe) -- at most 4 scalar branches instead of a 64-way switch.
#define SVD_VMCNT_IMM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ void svd_wait_vmcnt_coarse(int n) {
    if (n >= 8) {
        if (n >= 24) { if (n >= 48) SVD_VMCNT_IMM(48); else if (n >= 32) SVD_VMCNT_IMM(32); else SVD_VMCNT_IMM(24); }
        else { if (n >= 16) SVD_VMCNT_IMM(16); else if (n >= 12) SVD_VMCNT_IMM(12); else SVD_VMCNT_IMM(8); }
    } else if (n >= 4) {
        if (n >= 6) SVD_VMCNT_IMM(6); else SVD_VMCNT_IMM(4);
    } else {
        if (n >= 2) { if (n >= 3) SVD_VMCNT_IMM(3); else SVD_VMCNT_IMM(2); }
        else { if (n >= 1) SVD_VMCNT_IMM(1); else SVD_VMCNT_IMM(0); }
    }
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// host-side error plumbing (defined in api.hip)
extern "C" void svd_set_error_(const char* what, hipError_t e);
#define SVD_CHECK_LAUNCH(what)                                  \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) { svd_set_error_(what, e__); return SVD_ELAUNCH; } \
    } while (0)
