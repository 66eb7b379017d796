// Write-bandwidth microbenchmark for the GEMM epilogue's store shapes (gfx950).
// Each wave writes a 32-row x W-byte sub-block of a row-major [M, ld] bf16 matrix per "pass", like the GEMM epilogue:
//   mode 0: 16 rows x 64 B per store instruction (4 lanes x 16 B per row)  -- the current epilogue
//   mode 1: 8 rows x 128 B per store instruction (8 lanes per row)
//   mode 2: 4 rows x 256 B per instruction
//   mode 3: 1 KiB contiguous per instruction (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k(uint4* out, long M, int ldb /*bytes*/, int mode, int tiles_n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;          // 8 waves: 4 (m) x 2 (n), tile 256 rows x 256 B... 
    const long ntm = M / 256;
    const long ntiles = ntm * tiles_n;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long tm = t / tiles_n; const int tn = t % tiles_n;
        char* base = (char*)out + (tm * 256 + (wave >> 1) * 64) * (long)ldb + tn * 256 + (wave & 1) * 128;   // wave: 64 rows x 128 B
        if (mode == 0) {          // 2 fragments of 64 B width; per fragment 32 rows in 2 instr -> 64 rows: 4 instr per 64-B column, 8 total
            for (int c = 0; c < 2; ++c) for (int r = 0; r < 4; ++r)
                *(uint4*)(base + (long)(r * 16 + (lane >> 2)) * ldb + c * 64 + (lane & 3) * 16) = v;
        } else if (mode == 1) {   // 8 rows x 128 B
            for (int r = 0; r < 8; ++r) *(uint4*)(base + (long)(r * 8 + (lane >> 3)) * ldb + (lane & 7) * 16) = v;
        } else if (mode == 2) {   // waves arranged 8 (m) x 1: 32 rows x 256 B per wave; 4 rows x 256 B per instruction
            char* b2 = (char*)out + (tm * 256 + wave * 32) * (long)ldb + tn * 256;
            for (int r = 0; r < 8; ++r) *(uint4*)(b2 + (long)(r * 4 + (lane >> 4)) * ldb + (lane & 15) * 16) = v;
        } else {                  // contiguous
            char* b3 = (char*)out + (t * 8 + wave) * 8192L;
            for (int r = 0; r < 8; ++r) *(uint4*)(b3 + r * 1024 + lane * 16) = v;
        }
    }
}
int main() {
    const long M = 460800; const int N = 1280; const int ldb = N * 2; const int tiles_n = ldb / 256;
    uint4* d; hipMalloc(&d, M * (long)ldb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {256, 512, 1024}) for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(a); hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, d, M, ldb, mode, tiles_n); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("grid %4d mode %d: %.3f ms  %.2f TB/s\n", grid, mode, best, M * (double)ldb / best / 1e9);
    }
    return 0;
}
