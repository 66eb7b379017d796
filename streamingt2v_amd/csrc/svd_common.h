// Shared device helpers for the gfx950 kernels of libsvdhip.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/svdhip.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define SVD_WAVE 64

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
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
