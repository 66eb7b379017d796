// Access-pattern microbenchmark behind csrc/rowgemm.hip (round 6): the HBM traffic of one row-owning 320 -> 320 projection with its LayerNorm output --
// per 32-row tile: read X (16 bit, 20 KiB) + R (fp32, 40 KiB), write Y (fp32, 40 KiB) + Yn (16 bit, 20 KiB) of row-major [M, 320] tensors -- as a pure copy, in the
// lane <-> address patterns a kernel can give its memory instructions, at 4 / 8 / 16 waves per CU:
//   pattern 0  lane = ROW:      16 B per lane, the two half-waves side by side -> one instruction touches 32 rows x 32 B      (rowgemm v1: MFMA D^T fragments as they are)
//   pattern 1  4 lanes per row: 16 B per lane at a 32-byte pitch               -> 16 rows x 64 B per instruction              (gemm_impl.inc's LDS-staged epilogue)
//   pattern 2  8 lanes per row: 128 B contiguous per row                       -> 8 rows x 128 B per instruction
//   pattern 3  contiguous:      1 KiB per instruction                                                                       (upper bound: what a copy kernel does)
//   pattern 4  lane = CHANNEL:  4 B per lane (2 B for the 16-bit tensors), half-wave = 32 consecutive channels -> 2 rows x 128 B per instruction (MFMA D fragments untransposed)
// Build / run:  hipcc --offload-arch=gfx950 -O3 tools/access_pattern_bench.hip -o tools/_bin/access_pattern_bench && tools/_bin/access_pattern_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PAT>
__global__ __launch_bounds__(256) void k(const uint16_t* __restrict__ X, const float* __restrict__ R, float* __restrict__ Y, uint16_t* __restrict__ Yn, int ntiles) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += nw) {
        const char* xb = (const char*)X + (size_t)tile * 32 * 640;
        const char* rb = (const char*)R + (size_t)tile * 32 * 1280;
        char* yb = (char*)Y + (size_t)tile * 32 * 1280;
        char* nb = (char*)Yn + (size_t)tile * 32 * 640;
        if constexpr (PAT == 4) {
            float r[160]; uint16_t x[160];
            const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
            for (int o = 0; o < 10; ++o)
#pragma unroll
                for (int t = 0; t < 16; ++t) r[o * 16 + t] = *(const float*)(rb + (size_t)((t & 3) + 8 * (t >> 2) + 4 * hi) * 1280 + (32 * o + l31) * 4);
#pragma unroll
            for (int o = 0; o < 10; ++o)
#pragma unroll
                for (int t = 0; t < 16; ++t) x[o * 16 + t] = *(const uint16_t*)(xb + (size_t)((t & 3) + 8 * (t >> 2) + 4 * hi) * 640 + (64 * o + 2 * l31) * 1);   // 2 B per lane
#pragma unroll
            for (int o = 0; o < 10; ++o)
#pragma unroll
                for (int t = 0; t < 16; ++t) *(float*)(yb + (size_t)((t & 3) + 8 * (t >> 2) + 4 * hi) * 1280 + (32 * o + l31) * 4) = r[o * 16 + t] + 1.f;
#pragma unroll
            for (int o = 0; o < 10; ++o)
#pragma unroll
                for (int t = 0; t < 16; ++t) *(uint16_t*)(nb + (size_t)((t & 3) + 8 * (t >> 2) + 4 * hi) * 640 + (64 * o + 2 * l31) * 1) = x[o * 16 + t];
        } else {
            // byte offset of this lane's 16 B in instruction i of a [32 rows x ROWB bytes] block
            auto off = [&](int i, int rowb) -> size_t {
                if constexpr (PAT == 0) { const int per_row = rowb / 32; return (size_t)(lane & 31) * rowb + (i % per_row) * 32 + (lane >> 5) * 16; }
                else if constexpr (PAT == 1) { const int per_row = rowb / 128; const int rg = i / (2 * per_row), c = (i / 2) % per_row, h = i & 1;
                                               return (size_t)(rg * 16 + (lane >> 2)) * rowb + c * 128 + (lane & 3) * 32 + h * 16; }
                else if constexpr (PAT == 2) { const int per_row = rowb / 128; return (size_t)((i / per_row) * 8 + (lane >> 3)) * rowb + (i % per_row) * 128 + (lane & 7) * 16; }
                else { return (size_t)i * 1024 + lane * 16; }
            };
            uint4 r[40], x[20];
#pragma unroll
            for (int i = 0; i < 40; ++i) r[i] = *(const uint4*)(rb + off(i, 1280));
#pragma unroll
            for (int i = 0; i < 20; ++i) x[i] = *(const uint4*)(xb + off(i, 640));
#pragma unroll
            for (int i = 0; i < 40; ++i) { uint4 v = r[i]; v.x += 1; *(uint4*)(yb + off(i, 1280)) = v; }
#pragma unroll
            for (int i = 0; i < 20; ++i) *(uint4*)(nb + off(i, 640)) = x[i];
        }
    }
    if (threadIdx.x == 9999) smem[0] = 1;
}

template <int PAT> float run(const uint16_t* X, const float* R, float* Y, uint16_t* Yn, int ntiles, int wg_per_cu, int ncu) {
    // occupancy through dynamic LDS: 160 KiB / wg_per_cu each
    const int lds = wg_per_cu == 1 ? 96 * 1024 : (wg_per_cu == 2 ? 64 * 1024 : 40 * 1024);          // 96 KiB: one workgroup per CU; 64 KiB: two; 40 KiB: four
    hipError_t e = hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) { printf("pattern %d: hipFuncSetAttribute(%d): %s\n", PAT, lds, hipGetErrorString(e)); fflush(stdout); return -1.f; }
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<PAT>, dim3(ncu * wg_per_cu), dim3(256), lds, 0, X, R, Y, Yn, ntiles);
        hipEventRecord(b);
        e = hipEventSynchronize(b);
        if (e != hipSuccess || (e = hipGetLastError()) != hipSuccess) { printf("pattern %d wg %d: %s\n", PAT, wg_per_cu, hipGetErrorString(e)); fflush(stdout); return -1.f; }
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("  [pattern %d, %d workgroup(s) per CU: %.1f us]\n", PAT, wg_per_cu, best * 1e3); fflush(stdout);
    return best;
}

int main() {
    const long M = 460800; const int ntiles = M / 32;
    uint16_t *X, *Yn; float *R, *Y;
    hipMalloc(&X, M * 640); hipMalloc(&Yn, M * 640); hipMalloc(&R, M * 1280); hipMalloc(&Y, M * 1280);
    if (!X || !Yn || !R || !Y) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(X, 0, M * 640); hipMemset(R, 0, M * 1280);
    if (hipDeviceSynchronize() != hipSuccess) { printf("memset failed\n"); return 1; }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double bytes = (double)M * 3840;
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("# X 16-bit + R fp32 read, Y fp32 + Yn 16-bit written, [%ld, 320] row-major: %.2f GB per pass; %d CUs; 4 waves per workgroup\n", M, bytes / 1e9, p.multiProcessorCount);
    const char* names[5] = {"lane = row (32 rows x 32 B per instruction)", "4 lanes per row (16 rows x 64 B)", "8 lanes per row (8 rows x 128 B)", "contiguous 1 KiB", "lane = channel, 4 / 2 B per lane (2 rows x 128 / 64 B)"};
    for (int wg = 1; wg <= 4; wg *= 2) {
        float t[5] = {run<0>(X, R, Y, Yn, ntiles, wg, p.multiProcessorCount), run<1>(X, R, Y, Yn, ntiles, wg, p.multiProcessorCount), run<2>(X, R, Y, Yn, ntiles, wg, p.multiProcessorCount),
                      run<3>(X, R, Y, Yn, ntiles, wg, p.multiProcessorCount), run<4>(X, R, Y, Yn, ntiles, wg, p.multiProcessorCount)};
        for (int i = 0; i < 5; ++i) printf("%2d waves/CU  pattern %d  %-56s %8.1f us  %5.2f TB/s\n", 4 * wg, i, names[i], t[i] * 1e3, bytes / t[i] / 1e9);
    }
    return 0;
}
