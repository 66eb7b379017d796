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

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void svd_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
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
