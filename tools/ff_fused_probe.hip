// Standalone probe of the fused GEGLU feed-forward kernel (streamingt2v_amd/csrc/ff_fused.hip) against the product's two-launch path
// (svd_gemm GEGLU projection -> svd_gemm down-projection + fp32 residual) in ONE process on the same operands:
//     tools/_bin/ff_fused_probe [M = 460800] [reps = 20]
// Checks (1) fused vs two-launch output over every element, (2) both against a double-precision CPU evaluation of 192 sampled rows (operands
// as the kernels see them: fp16 X / W, fp32 biases, the hidden activation rounded to fp16), (3) times both, and the probe variants of the
// fused kernel (no LDS-DMA / no GELU / neither / no S^T MFMAs: wrong results, timing only) for a cost breakdown.
// Build: see tools/build_ff_probe.sh (compiles ff_fused.hip into this binary with -DSVD_FF_PROBES; the baseline comes from libsvdhip.so).
#define SVD_FF_PROBES 1
#define svd_ff_geglu_fused svd_ff_geglu_fused_probe
#define svd_ff_fused_pack_bytes svd_ff_fused_pack_bytes_probe
#include "../streamingt2v_amd/csrc/ff_fused.hip"
#undef svd_ff_geglu_fused
#undef svd_ff_fused_pack_bytes
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static inline uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
static inline float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

// matrix-pipe ceiling at the clock the chip actually runs: W waves per workgroup, one workgroup per CU, back-to-back MFMAs on registers only
template <int NACC>
__global__ __launch_bounds__(512, 1) void mfma_only_kernel(float* out, int iters) {
    f32x16_t acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = (float)(threadIdx.x + a);
    uint4 fa = make_uint4(threadIdx.x * 2654435761u, 0x3c003c00u, threadIdx.x, 0x3c003800u), fb = make_uint4(0x38003c00u, threadIdx.x * 40503u, 0x3c003c00u, 7u);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = ElemF16::mfma(fa, fb, acc[a]);
    }
    float sum = 0.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) sum += acc[a][0] + acc[a][15];
    if (sum == 123.456f) out[threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 460800;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int C = 320, HD = 1280;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> X((size_t)M * C), W1((size_t)2 * HD * C), W2((size_t)C * HD);
    std::vector<float> b1(2 * HD), b2(C), R((size_t)M * C);
    for (auto& v : X) v = f2h(nd(rng));
    for (auto& v : W1) v = f2h(0.06f * nd(rng));
    for (auto& v : W2) v = f2h(0.04f * nd(rng));
    for (auto& v : b1) v = 0.2f * nd(rng);
    for (auto& v : b2) v = 0.2f * nd(rng);
    for (auto& v : R) v = nd(rng);
    // ---- packed image of the fused kernel (same layout as streamingt2v_amd/video_model.pack_ff_fused)
    const int nch = HD / 32;
    std::vector<uint8_t> img((size_t)nch * FF_BLOB, 0);
    for (int c = 0; c < nch; ++c) {
        uint8_t* blob = img.data() + (size_t)c * FF_BLOB;
        for (int s = 0; s < FF_NS; ++s)
            for (int t = 0; t < 2; ++t)
                for (int l = 0; l < 64; ++l) {
                    const int m = l & 31, kg = l >> 5;
                    const int hid = 32 * c + 16 * t + (m & 15);
                    const int row = m < 16 ? hid : HD + hid;
                    uint16_t* d = (uint16_t*)(blob + (2 * s + t) * 1024 + l * 16);
                    for (int e = 0; e < 8; ++e) d[e] = W1[(size_t)row * C + 16 * s + 8 * kg + e];
                }
        float* bp = (float*)(blob + 2 * FF_NS * 1024);
        for (int t = 0; t < 2; ++t)
            for (int m = 0; m < 32; ++m) {
                const int hid = 32 * c + 16 * t + (m & 15);
                bp[t * 32 + m] = b1[m < 16 ? hid : HD + hid];
            }
        for (int t = 0; t < 2; ++t)
            for (int o = 0; o < FF_NO; ++o)
                for (int l = 0; l < 64; ++l) {
                    const int m = l & 31, kg = l >> 5;
                    uint16_t* d = (uint16_t*)(blob + FF_W1_BYTES + (t * FF_NO + o) * 1024 + l * 16);
                    for (int e = 0; e < 8; ++e) {
                        const int u = e < 4 ? 4 * kg + e : 8 + 4 * kg + (e - 4);
                        d[e] = W2[(size_t)(32 * o + m) * HD + 32 * c + 16 * t + u];
                    }
                }
    }
    // ---- the two-launch path's GEGLU weight: value | gate rows interleaved in blocks of 32 (video_model.pack_geglu)
    std::vector<uint16_t> W1i((size_t)2 * HD * C);
    std::vector<float> b1i(2 * HD);
    for (int blk = 0; blk < HD / 32; ++blk)
        for (int r = 0; r < 32; ++r) {
            memcpy(&W1i[(size_t)(blk * 64 + r) * C], &W1[(size_t)(blk * 32 + r) * C], C * 2);
            memcpy(&W1i[(size_t)(blk * 64 + 32 + r) * C], &W1[(size_t)(HD + blk * 32 + r) * C], C * 2);
            b1i[blk * 64 + r] = b1[blk * 32 + r];
            b1i[blk * 64 + 32 + r] = b1[HD + blk * 32 + r];
        }
    void *dX, *dImg, *dW1i, *dW2, *dH, *dZero;
    float *db1i, *db2, *dR, *dYf, *dYb;
    CK(hipMalloc(&dX, X.size() * 2)); CK(hipMalloc(&dImg, img.size())); CK(hipMalloc(&dW1i, W1i.size() * 2)); CK(hipMalloc(&dW2, W2.size() * 2));
    CK(hipMalloc(&dH, (size_t)M * HD * 2)); CK(hipMalloc(&dZero, 4096)); CK(hipMemset(dZero, 0, 4096));
    CK(hipMalloc(&db1i, b1i.size() * 4)); CK(hipMalloc(&db2, b2.size() * 4)); CK(hipMalloc(&dR, R.size() * 4));
    CK(hipMalloc(&dYf, R.size() * 4)); CK(hipMalloc(&dYb, R.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dImg, img.data(), img.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dW1i, W1i.data(), W1i.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, W2.data(), W2.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db1i, b1i.data(), b1i.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dR, R.data(), R.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dYf, 0xff, R.size() * 4)); CK(hipMemset(dYb, 0xff, R.size() * 4));

    auto fused = [&]() {
        const int rc = svd_ff_geglu_fused_probe((const svd_bf16*)dX, C, dImg, C, HD, db2, dR, C, nullptr, 0, 0.f, 1, dYf, C, 1, M, SVD_DTYPE_F16, nullptr);
        if (rc) { printf("svd_ff_geglu_fused rc %d: %s\n", rc, svd_last_error()); exit(3); }
    };
    auto baseline = [&]() {
        svd_gemm_args a;
        memset(&a, 0, sizeof a);
        a.A = (const svd_bf16*)dX; a.lda = C; a.W = (const svd_bf16*)dW1i; a.ldw = C; a.M = (int)M; a.N = 2 * HD; a.K = C; a.a_mode = SVD_A_PLAIN;
        a.zeros = (const svd_bf16*)dZero; a.bias = db1i; a.epi_flags = SVD_EPI_GEGLU; a.C = dH; a.ldc = HD; a.out_mode = SVD_OUT_BF16; a.dtype = SVD_DTYPE_F16;
        int rc = svd_gemm(&a, nullptr);
        if (rc) { printf("svd_gemm(geglu) rc %d: %s\n", rc, svd_last_error()); exit(3); }
        memset(&a, 0, sizeof a);
        a.A = (const svd_bf16*)dH; a.lda = HD; a.W = (const svd_bf16*)dW2; a.ldw = HD; a.M = (int)M; a.N = C; a.K = HD; a.a_mode = SVD_A_PLAIN;
        a.zeros = (const svd_bf16*)dZero; a.bias = db2; a.R = (const svd_bf16*)dR; a.ldr = C; a.res_f32 = 1; a.C = dYb; a.ldc = C; a.out_mode = SVD_OUT_F32;
        a.dtype = SVD_DTYPE_F16;
        rc = svd_gemm(&a, nullptr);
        if (rc) { printf("svd_gemm(ff2) rc %d: %s\n", rc, svd_last_error()); exit(3); }
    };
    // the four-wave form first (probe variant -1), kept for the bit comparison with the eight-wave default
    std::vector<float> Y4(R.size());
    svd_ff_probe_variant = -1;
    fused();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(Y4.data(), dYf, Y4.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dYf, 0xff, R.size() * 4));
    svd_ff_probe_variant = 0;
    fused(); baseline();
    CK(hipDeviceSynchronize());
    std::vector<float> Yf(R.size()), Yb(R.size());
    CK(hipMemcpy(Yf.data(), dYf, Yf.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Yb.data(), dYb, Yb.size() * 4, hipMemcpyDeviceToHost));
    {
        size_t ndiff = 0, first = 0;
        for (size_t i = 0; i < Yf.size(); ++i)
            if (memcmp(&Yf[i], &Y4[i], 4) != 0) { if (!ndiff) first = i; ++ndiff; }
        printf("[eight-wave vs four-wave fused kernel, all %zu elements] differing bit patterns: %zu%s\n", Yf.size(), ndiff, ndiff ? "" : "  (bit-identical)");
        if (ndiff) printf("   first difference at row %zu channel %zu: %.9g vs %.9g\n", first / C, first % C, Yf[first], Y4[first]);
    }
    double maxd = 0, sumsq = 0; size_t nbad = 0, nnan = 0;
    for (size_t i = 0; i < Yf.size(); ++i) {
        if (!(Yf[i] == Yf[i])) { ++nnan; continue; }
        const double d = fabs((double)Yf[i] - Yb[i]);
        maxd = d > maxd ? d : maxd; sumsq += d * d;
        if (d > 2e-3) ++nbad;
    }
    printf("[fused vs two-launch, all %zu elements] max |diff| %.3e  rms diff %.3e  elements off by > 2e-3: %zu  NaN/unwritten: %zu\n", Yf.size(), maxd,
           sqrt(sumsq / Yf.size()), nbad, nnan);
    // CPU double reference on sampled rows
    double ef = 0, eb = 0, refsq = 0; size_t nref = 0;
    for (int k = 0; k < 192; ++k) {
        const int64_t row = k < 64 ? k * 2 + (k & 1) * 31 : (k < 128 ? M - 1 - (k - 64) * 3 : (int64_t)((double)(k - 128) / 64.0 * (M - 1)));
        std::vector<double> h(HD);
        for (int u = 0; u < HD; ++u) {
            double v = b1[u], g = b1[HD + u];
            for (int c = 0; c < C; ++c) {
                const double x = h2f(X[(size_t)row * C + c]);
                v += x * h2f(W1[(size_t)u * C + c]); g += x * h2f(W1[(size_t)(HD + u) * C + c]);
            }
            const double hv = v * 0.5 * g * (1.0 + erf(g * 0.70710678118654752));
            h[u] = h2f(f2h((float)hv));
        }
        for (int c = 0; c < C; ++c) {
            double o = b2[c] + (double)R[(size_t)row * C + c];
            for (int u = 0; u < HD; ++u) o += h[u] * h2f(W2[(size_t)c * HD + u]);
            const double df = fabs(o - Yf[(size_t)row * C + c]), dbb = fabs(o - Yb[(size_t)row * C + c]);
            ef = df > ef ? df : ef; eb = dbb > eb ? dbb : eb; refsq += o * o; ++nref;
        }
    }
    printf("[vs CPU double, 192 rows] max |err| fused %.3e  two-launch %.3e  (output rms %.3f)\n", ef, eb, sqrt(refsq / nref));
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, const char* name) {
        for (int i = 0; i < 5; ++i) fn();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double fl = 2.0 * M * (2.0 * HD) * C + 2.0 * M * HD * C;
        printf("%-44s %8.3f ms   %7.1f TFLOP/s\n", name, ms, fl / ms * 1e-9);
        return ms;
    };
    timeit(fused, "fused feed-forward, 8 waves (1 launch)");
    timeit(baseline, "two launches (GEGLU proj + down-proj)");
    timeit(fused, "fused feed-forward, 8 waves (again)");
    svd_ff_probe_variant = -1;
    timeit(fused, "fused feed-forward, 4 waves (round-5 first form)");
    svd_ff_probe_variant = 0;
    svd_ff_probe_variant = -1;
    for (int n : {0, 1, 2, 3, 4, 6}) {
        CK(hipMemcpyToSymbol(HIP_SYMBOL(ff_dephase_probe), &n, sizeof(int)));
        char name[96];
        snprintf(name, sizeof name, "4 waves, start de-phasing: %d x s_sleep 127 per slot", n);
        timeit(fused, name);
    }
    { int n = -1; CK(hipMemcpyToSymbol(HIP_SYMBOL(ff_dephase_probe), &n, sizeof(int))); }
    svd_ff_probe_variant = 0;
    for (int waves : {4, 8}) {
        const int iters = 20000;
        float* dO; CK(hipMalloc(&dO, 4096));
        hipLaunchKernelGGL(mfma_only_kernel<4>, dim3(256), dim3(waves * 64), 0, 0, dO, 100);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_only_kernel<4>, dim3(256), dim3(waves * 64), 0, 0, dO, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 256.0 * waves * iters * 4.0 * 32768.0;
        printf("matrix-pipe ceiling: %d waves / CU, registers only, 32x32x16 f16   %8.3f ms   %7.1f TFLOP/s\n", waves, ms, fl / ms * 1e-9);
    }
    {
        const int vs[8] = {116, 101, 102, 103, 104, 108, 107, 115};
        const char* ns[8] = {"8w probe: fragments 7 MFMAs ahead (CORRECT results)", "8w probe: no LDS-DMA in the steps", "8w probe: no GELU arithmetic", "8w probe: no DMA, no GELU",
                             "8w probe: no workgroup barrier in the steps", "8w probe: no fragment reads in the steps", "8w probe: no DMA, GELU, barrier", "8w probe: no DMA, GELU, barrier, fragment reads"};
        for (int k = 0; k < 8; ++k) { svd_ff_probe_variant = vs[k]; timeit(fused, ns[k]); }
        svd_ff_probe_variant = 0;
    }
    if (argc > 3) return (nnan || nbad > Yf.size() / 100000 + 10 || ef > 5e-3) ? 1 : 0;
    const char* names[5] = {"", "probe: no LDS-DMA in the steps", "probe: no GELU arithmetic", "probe: no DMA, no GELU", "probe: no S^T MFMAs"};
    for (int v = 1; v <= 4; ++v) { svd_ff_probe_variant = v; timeit(fused, names[v]); }
    svd_ff_probe_variant = 0;
    return (nnan || nbad > Yf.size() / 100000 + 10 || ef > 5e-3) ? 1 : 0;
}
