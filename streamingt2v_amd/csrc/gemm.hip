// Dispatcher of the GEMM family: argument validation, tile heuristic, per-element-type launch tables.
// Kernel: gemm_impl.inc (instantiated for bf16 in gemm_bf16.hip and fp16 in gemm_f16.hip); tiles: gemm_cfg.h.
#include "gemm_cfg.h"

using namespace svd_gemm_detail;

namespace {

template <class CFG>
void info(int* bm, int* bn, int* thr, int* lds) {
    if (bm) *bm = CFG::BM; if (bn) *bn = CFG::BN; if (thr) *thr = CFG::THREADS; if (lds) *lds = CFG::TOTAL_LDS;
}

// can config `cfg` run these arguments?  (tile-shape independent checks live in svd_gemm)
template <class CFG>
bool cfg_ok(const svd_gemm_args& a) {
    const int kq = (a.a_mode == SVD_A_PLAIN) ? a.K : a.cin;
    if (kq % CFG::BK != 0 || a.K % CFG::BK != 0) return false;
    if (CFG::TRANS != (a.out_mode == SVD_OUT_BF16_T)) return false;
    if ((a.epi_flags & SVD_EPI_GEGLU) && (CFG::FN % 2 != 0)) return false;   // value|gate frag pairs per wave
    if (a.a_mode == SVD_A_CONV3X3 && a.ups && !CFG::UPS_KERNEL) return false; // the folded-upsample kernel exists for a subset of the tiles
    if ((a.res_f32 || a.out_mode == SVD_OUT_F32) && !CFG::STREAM_KERNEL) return false;   // likewise the fp32-residual-stream kernel
    return true;
}

int pick_cfg(const svd_gemm_args& a) {
    const int kq = (a.a_mode == SVD_A_PLAIN) ? a.K : a.cin;
    if (a.out_mode == SVD_OUT_BF16_T) return 7;
    if (kq % 64 != 0 || a.K % 64 != 0) return 5;
    if (a.epi_flags & SVD_EPI_GEGLU) return (a.M >= 65536) ? 2 : 1;
    if (a.N % 320 == 0 && a.N <= 960) return 4;
    if (a.N <= 64) return 3;
    if (a.M >= 65536 && a.N % 128 == 0) return 2;
    return 1;
}

}  // namespace

extern "C" int svd_gemm_num_configs(void) { return kNumCfg; }
extern "C" int svd_gemm_pick_config(const svd_gemm_args* args) { return args ? pick_cfg(*args) : SVD_EINVAL; }

extern "C" int svd_gemm_config_info(int cfg, int* bm, int* bn, int* threads, int* lds_bytes) {
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: info<Cfg##id>(bm, bn, threads, lds_bytes); return SVD_OK;
        SVD_GEMM_CONFIGS(X)
#undef X
    }
    return SVD_EINVAL;
}

extern "C" int svd_gemm_config_valid(const svd_gemm_args* args, int cfg) {
    if (!args) return SVD_EINVAL;
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: return cfg_ok<Cfg##id>(*args) ? 1 : 0;
        SVD_GEMM_CONFIGS(X)
#undef X
    }
    return SVD_EINVAL;
}

namespace {
// the tile table is instantiated in four parts per element type (gemm_cfg.h)
int dispatch(const svd_gemm_args& a, int cfg, hipStream_t s, int m_base) {
    const bool f16 = a.dtype == SVD_DTYPE_F16;
    if (!f16 && a.dtype != SVD_DTYPE_BF16) return SVD_EINVAL;
    switch (cfg) {
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: return f16 ? svd_gemm_launch_f16_p0(a, cfg, s, m_base) : svd_gemm_launch_bf16_p0(a, cfg, s, m_base);
        SVD_GEMM_CONFIGS_P0(X)
#undef X
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: return f16 ? svd_gemm_launch_f16_p1(a, cfg, s, m_base) : svd_gemm_launch_bf16_p1(a, cfg, s, m_base);
        SVD_GEMM_CONFIGS_P1(X)
#undef X
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: return f16 ? svd_gemm_launch_f16_p2(a, cfg, s, m_base) : svd_gemm_launch_bf16_p2(a, cfg, s, m_base);
        SVD_GEMM_CONFIGS_P2(X)
#undef X
#define X(id, bm_, bn_, wm, wn, bk, glds, tr, ns) case id: return f16 ? svd_gemm_launch_f16_p3(a, cfg, s, m_base) : svd_gemm_launch_bf16_p3(a, cfg, s, m_base);
        SVD_GEMM_CONFIGS_P3(X)
#undef X
    }
    return SVD_EINVAL;
}
}  // namespace

// The tail of a launch (rows [m_base, a.M), gemm_impl.inc launch_kernel) with the smallest tile that carries these arguments: 128 x 128 first (cfg 1; BK = 32: cfg 5),
// then 128-row tiles of the widths the stream / upsample kernels exist for.  SVD_EINVAL = none (the caller finishes the rows with its own tile).
int svd_gemm_tail_launch_(const svd_gemm_args& a, int main_bm, int m_base, hipStream_t s) {
    static const int prefer[] = {1, 5, 22, 4, 3};
    for (int cfg : prefer) {
        int bm = 0, bn = 0, th = 0, lds = 0;
        if (svd_gemm_config_info(cfg, &bm, &bn, &th, &lds) != SVD_OK || bm >= main_bm || m_base % bm) continue;
        if (svd_gemm_config_valid(&a, cfg) != 1) continue;
        return dispatch(a, cfg, s, m_base);
    }
    return SVD_EINVAL;
}

extern "C" int svd_gemm(const svd_gemm_args* args, svd_stream_t stream) {
    if (!args) return SVD_EINVAL;
    const svd_gemm_args& a = *args;
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return SVD_EINVAL;
    if (a.N % 4 != 0 && a.out_mode != SVD_OUT_BF16_T) return SVD_EINVAL;
    if (a.lda % 8 != 0 || a.ldw % 8 != 0) return SVD_EINVAL;             // 16-byte staged loads
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return SVD_EINVAL;
    if (a.a_mode != SVD_A_PLAIN) {
        if (!a.zeros || a.cin <= 0 || a.cin % 32 != 0) return SVD_EINVAL;
        if (a.a_mode == SVD_A_CONV3X3) {
            if (a.K != 9 * a.cin || a.hin <= 0 || a.win <= 0 || a.hout <= 0 || a.wout <= 0) return SVD_EINVAL;
            if ((a.stride != 1 && a.stride != 2) || (a.ups != 0 && a.ups != 1)) return SVD_EINVAL;
            if (a.M % (a.hout * a.wout) != 0) return SVD_EINVAL;
            if (a.pad_mode != 0 && (a.pad_mode != 1 || a.stride != 2)) return SVD_EINVAL;
            if (a.ups && (a.stride != 1 || a.hout > 2 * a.hin || a.hout < 2 * a.hin - 1 || a.wout > 2 * a.win || a.wout < 2 * a.win - 1))
                return SVD_EINVAL;
        } else if (a.a_mode == SVD_A_TEMPORAL3) {
            if (a.K != 3 * a.cin || a.t_frames <= 0 || a.rows_per_frame <= 0) return SVD_EINVAL;
            if (a.M % (a.t_frames * a.rows_per_frame) != 0) return SVD_EINVAL;
        } else return SVD_EINVAL;
    }
    if (a.K % 32 != 0) return SVD_EINVAL;
    if (a.a_mode != SVD_A_PLAIN) {   // the implicit-GEMM views address the source with 32-bit ELEMENT offsets (gemm_impl.inc a_off / pix * lda): reject sources they cannot reach
        int64_t src_rows = a.M;
        if (a.a_mode == SVD_A_CONV3X3) src_rows = (int64_t)(a.M / (a.hout * a.wout)) * a.hin * a.win + a.win + 2;   // + one row and column: offsets are taken from tap (1, 1)
        if (src_rows * a.lda >= ((int64_t)1 << 32)) return SVD_EINVAL;
    } else {                         // plain A: 64-bit wave-uniform tile base + a 32-bit per-lane BYTE offset inside the tile's rows (at most 320 rows)
        if ((int64_t)320 * a.lda * 2 >= ((int64_t)1 << 32)) return SVD_EINVAL;
    }
    if ((int64_t)320 * a.ldw * 2 >= ((int64_t)1 << 32)) return SVD_EINVAL;      // W tiles: same SADDR form
    if (a.rowvec && a.rows_per_vec <= 0) return SVD_EINVAL;
    if (a.out_mode == SVD_OUT_BF16_T) {
        if (a.tok_per_frame <= 0 || a.tok_per_frame % 4 != 0 || a.M % a.tok_per_frame != 0 || a.tokens_ld % 4 != 0)
            return SVD_EINVAL;
        if (a.R || a.S || a.rowvec || a.epi_flags) return SVD_EINVAL;
    }
    if (a.epi_flags & SVD_EPI_GEGLU) {
        // GEGLU projections write 16 bit and take no fp32 residual: the fp32-stream kernel has no GEGLU pass (it would store N/2-wide rows as N columns)
        if (a.N % 64 != 0 || a.out_mode != SVD_OUT_BF16 || a.res_f32) return SVD_EINVAL;
    }
    int cfg = a.tile_cfg > 0 ? a.tile_cfg : pick_cfg(a);
    if (a.tile_cfg > 0 && ((a.a_mode == SVD_A_CONV3X3 && a.ups) || a.res_f32 || a.out_mode == SVD_OUT_F32) && svd_gemm_config_valid(args, cfg) != 1)
        cfg = pick_cfg(a);   // a tile of a tuned table that does not carry the folded-upsample / fp32-stream kernel: the heuristic's pick does
    if (svd_gemm_config_valid(args, cfg) != 1) return SVD_EINVAL;
    return dispatch(a, cfg, (hipStream_t)stream, 0);
}
