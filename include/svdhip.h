/*
 * svdhip.h -- C ABI of libsvdhip.so: the MI355X (gfx950 / CDNA4) kernels behind the StreamingSVD
 * denoising hot path (StreamingWrapper.forward -> ControlNet + VideoUNet(+CAM), Euler-EDM step glue,
 * temporal VAE decode).
 *
 * The reference (Picsart-AI-Research/StreamingT2V) has no FFI: its "operators" are the PyTorch/xformers/cuDNN
 * library ops its nn.Modules dispatch to.  Every entry point below replaces one such
 * library dispatch; the reference call site is cited per function (paths relative to /root/reference/code).
 *
 * Contract (all entry points):
 *   - plain pointers + sizes, no torch types; the caller owns every buffer (device memory) and the stream;
 *   - stateless, re-entrant, asynchronous on `stream`; no allocation, no host sync inside;
 *   - returns 0 on success, a negative SVD_E* code on bad arguments / launch failure (no exceptions);
 *   - activations are "channels-last token" tensors: row m = (frame, pixel), contiguous channels,
 *     explicit row stride (`ld*`, in elements).  16-bit storage (fp16 or bf16, per call: `dtype`), fp32 accumulation everywhere.
 */
#ifndef SVDHIP_H
#define SVDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* svd_stream_t;      /* hipStream_t */
typedef uint16_t svd_bf16;       /* raw 16-bit element (bf16 or fp16 bits, see `dtype`) */

/* element type of every 16-bit tensor of a call: fp16 -- the reference's own autocast precision (config.yaml:8 "16-mixed"), the host package's
 * default (ops.DEFAULT_ELEM) and the type every parity bound and bench line is stated in -- or bf16 (what north_star names; same MFMA rate, 8x
 * coarser rounding; selectable, ops.set_element_dtype).  Storage 16 bit, accumulation / statistics fp32 in both.  SVD_DTYPE_F32 only where stated. */
enum { SVD_DTYPE_BF16 = 0, SVD_DTYPE_F16 = 1, SVD_DTYPE_F32 = 2 };
/* OR-ed into the 16-bit `dtype` of the READERS of the fp32 residual stream (svd_groupnorm_stats / _sums / _apply, svd_layernorm,
 * svd_add_rows): the input X (and svd_layernorm's Xsum / svd_add_rows' Y, which continue the stream) is fp32 with leading dimensions in
 * fp32 elements; the normalised output Y -- a GEMM operand -- stays in the 16-bit type.  The stream option keeps the tensors the
 * reference's residual additions run on (models/svd/sgm/modules/diffusionmodules/openaimodel.py:351-354, attention.py:567-593,
 * video_attention.py:125-168) in fp32 between the kernels: only what a matrix core consumes is rounded to 16 bit. */
enum { SVD_DTYPE_IN_F32 = 0x100 };

enum {
    SVD_OK = 0,
    SVD_EINVAL = -1,   /* bad shape / alignment / unsupported combination */
    SVD_ELAUNCH = -2,  /* hip launch error */
};

/* ---- library ------------------------------------------------------------------------------------------- */
/* ABI version of this header; bumped on any signature change. */
int svd_abi_version(void);
/* hipGetLastError()-style text of the last launch failure on this thread ("" if none). */
const char* svd_last_error(void);

/* ---- GEMM family (MFMA bf16, fp32 accumulate) ------------------------------------------------------------
 * C[M,N] = epilogue( A_view[M,K] . W[N,K]^T )
 * Replaces: every nn.Linear (cuBLAS addmm; models/svd/sgm/modules/attention.py:94-120,262-351),
 *           nn.Conv2d 3x3 / 1x1 (cuDNN; models/svd/sgm/modules/diffusionmodules/openaimodel.py:257-314,134-137,192-199),
 *           nn.Conv3d (3,1,1) of VideoResBlock.time_stack (models/diffusion/video_model.py:46-59),
 *           F.interpolate(nearest, 2x) feeding Upsample.conv (openaimodel.py:139-157).
 * A_view modes (implicit GEMM, nothing is materialised):
 *   SVD_A_PLAIN    : row m of A, K contiguous.
 *   SVD_A_CONV3X3  : m = (f, yo, xo); K = 9*cin ordered (ky, kx, c); zero padding 1; stride 1|2; optional
 *                    nearest 2x upsample of the source (ups=1; hout in {2 hin - 1, 2 hin}) folded into the addressing.
 *   SVD_A_TEMPORAL3: m = (b, t, p); K = 3*cin ordered (kt, c); zero padding at t=0 and t=T-1.
 * Epilogue, in this order (each optional):  v = acc + bias[n] + rowvec[m / rows_per_vec][n] + R[m][n];
 *   GEGLU (W rows interleaved in blocks of 32: value|gate): v = v_value * gelu_erf(v_gate), N_out = N/2;
 *   blend: v = alpha * S[m][n] + (1 - alpha) * v;   SiLU: v = silu(v).
 * Output: bf16 row-major (ldc), or fp32 row-major, or bf16 "transposed per frame" C[f][n][tok] (ld = tokens_ld)
 *   used to hand V^T to the attention kernels.
 */
enum { SVD_A_PLAIN = 0, SVD_A_CONV3X3 = 1, SVD_A_TEMPORAL3 = 2 };
enum { SVD_OUT_BF16 = 0, SVD_OUT_F32 = 1, SVD_OUT_BF16_T = 2 };
enum { SVD_EPI_GEGLU = 1, SVD_EPI_SILU = 2 /* v = silu(v) last (ControlNet cond-embedding conv_in, controlnet.py:105-106) */ };

typedef struct svd_gemm_args {
    /* operands */
    const svd_bf16* A;  int64_t lda;      /* activation rows, stride in elements */
    const svd_bf16* W;  int64_t ldw;      /* weights [N][K] row-major */
    int32_t M, N, K;
    /* A view */
    int32_t a_mode;
    int32_t cin;                          /* conv / temporal: channels per tap (K = taps*cin) */
    int32_t hin, win;                     /* conv: source frame size (before optional upsample) */
    int32_t hout, wout;                   /* conv: output frame size; M = frames*hout*wout */
    int32_t stride, ups;                  /* conv: stride 1|2 ; ups 0|1 (nearest 2x upsample of source) */
    int32_t t_frames, rows_per_frame;     /* temporal: T and pixels per frame */
    const svd_bf16* zeros;                /* >= 128 B of zeros (source of padded taps) */
    /* epilogue */
    const float* bias;                    /* [N] or NULL */
    const float* rowvec; int32_t rowvec_ld; int32_t rows_per_vec;  /* per-frame vector add, or NULL */
    const svd_bf16* R;  int64_t ldr;      /* residual, or NULL */
    const svd_bf16* S;  int64_t lds;      /* blend partner, or NULL */
    float alpha;
    int32_t epi_flags;
    /* output */
    void* C; int64_t ldc;
    int32_t out_mode;
    int32_t tok_per_frame; int64_t tokens_ld;  /* SVD_OUT_BF16_T only */
    /* tuning: 0 = heuristic, else explicit tile config id (see svd_gemm_num_configs) */
    int32_t tile_cfg;
    int32_t dtype;                        /* SVD_DTYPE_BF16 | SVD_DTYPE_F16 : A, W, R, S and 16-bit outputs */
    uint64_t* dbg_cycles;                 /* optional (NULL in production): 8 x u64 phase cycle counters of block 0 / wave 0 */
    int32_t pad_mode;                     /* conv, stride 2: 0 = zero padding 1 on every side; 1 = F.pad (0,1,0,1) then padding 0
                                             (Downsample of the VAE encoder, sgm/modules/diffusionmodules/model.py:73-92) */
    int32_t res_f32;                      /* 1: R and S are fp32 (the fp32 residual stream; ldr / lds in fp32 elements), 0: 16-bit `dtype` */
} svd_gemm_args;

int svd_gemm(const svd_gemm_args* args, svd_stream_t stream);
int svd_gemm_num_configs(void);
/* tile config the heuristic would choose for these arguments (1-based), or SVD_EINVAL */
int svd_gemm_pick_config(const svd_gemm_args* args);
/* 1 if tile config `cfg` can run these arguments, 0 if not, SVD_EINVAL for an unknown id (used by the autotuner) */
int svd_gemm_config_valid(const svd_gemm_args* args, int cfg);
/* fills bm/bn of config id (1-based); returns 0 or SVD_EINVAL */
int svd_gemm_config_info(int cfg, int* bm, int* bn, int* threads, int* lds_bytes);

/* ---- attention ------------------------------------------------------------------------------------------
 * Spatial self-attention, head dim 64, flash-style (scores never materialised).
 * Replaces xformers.memory_efficient_attention / F.scaled_dot_product_attention in
 * models/svd/sgm/modules/attention.py:339-343,434-441 (BasicTransformerBlock.attn1).
 *   Q,K : [frames][n_tok][heads][64] with row stride ldq/ldk (elements) between tokens, frame stride = n_tok*ld
 *   Vt  : [frames][heads*64][tok_ld]   (V transposed per frame; produced by svd_gemm SVD_OUT_BF16_T)
 *   O   : [frames][n_tok][heads][64], row stride ldo
 * scale = 1/sqrt(64) applied inside.
 */
int svd_attn_spatial_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                         const svd_bf16* Vt, int64_t tok_ld, svd_bf16* O, int64_t ldo,
                         int32_t frames, int32_t n_tok, int32_t heads, int32_t dtype, svd_stream_t stream);

/* Cross-attention form of the kernel above: n_q queries per frame attend to the n_k keys/values of key-value set
 * (frame / frames_per_kv) -- K rows [(frames/frames_per_kv) * n_k], Vt [frames/frames_per_kv][heads*64][tok_ld >= 64-multiple of n_k].
 * Replaces attn2 of the enhancer's Transformer2DModel blocks (code/i2v_enhance/attention.py:488-501: 77 text + 64 image-latent
 * + 4 CLIP tokens, shared by all frames of a batch element, unet_i2vgen_xl.py:664-690) -- diffusers AttnProcessor2_0 SDPA.
 * svd_attn_spatial_d64 == svd_attn_cross_d64 with n_q == n_k, frames_per_kv == 1. */
int svd_attn_cross_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                       const svd_bf16* Vt, int64_t tok_ld, svd_bf16* O, int64_t ldo,
                       int32_t frames, int32_t n_q, int32_t n_k, int32_t frames_per_kv, int32_t heads,
                       int32_t dtype, svd_stream_t stream);

/* Per-pixel temporal attention over short sequences (<= 128 frames: 32 -> half-waves, 64 -> waves, 128 -> waves with two query passes), head dim 64.
 * Also the two self-attentions of the enhancer's TransformerTemporalModel (code/i2v_enhance/transformer_temporal.py:160-195,
 * 38-frame chunks).
 * Replaces the attention inside VideoTransformerBlock.attn1 (models/svd/sgm/modules/video_attention.py:145-148,
 * on the "(b t) s c -> (b s) t c" view, never materialised here) and the diffusers Attention of the CAM
 * merger (models/cam/conditioning.py:65-68; q 25 frames, kv 7 ControlNet frames).
 *   Q : rows (b, tq, p) -> Q + ((b*Tq + tq)*n_pix + p)*ldq ; K,V likewise with Tk ; O like Q.
 */
int svd_attn_temporal_d64(const svd_bf16* Q, int64_t ldq, const svd_bf16* K, int64_t ldk,
                          const svd_bf16* V, int64_t ldv, svd_bf16* O, int64_t ldo,
                          int32_t batch, int32_t tq, int32_t tk, int32_t n_pix, int32_t heads,
                          int32_t dtype, svd_stream_t stream);

/* Row softmax: P[r][0..n) = softmax(scale * S[r][0..n)) ; fp32 in, bf16 out (VAE AttnBlock, model.py:180-201). */
int svd_softmax_rows(const float* S, int64_t lds, svd_bf16* P, int64_t ldp, int64_t rows, int32_t n,
                     float scale, int32_t dtype, svd_stream_t stream);

/* ---- normalisation --------------------------------------------------------------------------------------
 * GroupNorm(32 groups) on channels-last data, statistics in fp32 over (frames_per_stat frames x pixels x C/G).
 * Replaces GroupNorm32 / nn.GroupNorm (+ the following SiLU) of
 * models/svd/sgm/modules/diffusionmodules/util.py:259-276, openaimodel.py:257-261,292-305, attention.py:132-135,
 * models/cam/conditioning.py:33-34,57-59 (5-D statistics: frames_per_stat = T).
 *   stats: workspace, float [frames/frames_per_stat][groups][2] (mean, rstd) written by svd_groupnorm_stats.
 *   partial: workspace float, >= svd_groupnorm_partial_elems(...) elements.
 */
int64_t svd_groupnorm_partial_elems(int32_t frames, int32_t channels);
int svd_groupnorm_stats(const void* X, int64_t ldx, int32_t frames, int32_t pix, int32_t channels,
                        int32_t groups, int32_t frames_per_stat, float eps, float* partial, float* stats,
                        int32_t dtype, svd_stream_t stream);
int svd_groupnorm_apply(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t pix,
                        int32_t channels, int32_t groups, int32_t frames_per_stat, const float* stats,
                        const float* gamma, const float* beta, int32_t silu, int32_t dtype, svd_stream_t stream);
/* One-call form (ABI v7): statistics pass + apply pass.  When a (stat batch, group) pair has <= 64 partial sums (every per-frame norm of
 * the UNet / ControlNet: util.py:274-276 GroupNorm32 of the 2-D ResBlocks, attention.py:132-135) the apply pass reduces them itself and no
 * finalize kernel runs (`stats` untouched); otherwise identical to svd_groupnorm_stats followed by svd_groupnorm_apply. */
int svd_groupnorm(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t pix, int32_t channels,
                  int32_t groups, int32_t frames_per_stat, float eps, float* partial, float* stats, const float* gamma,
                  const float* beta, int32_t silu, int32_t dtype, svd_stream_t stream);
/* Sequence-parallel form of the 5-D statistics (time_stack GroupNorms models/diffusion/video_model.py:75-80 and the CAM merger's
 * norm models/cam/conditioning.py:57-59 pool over ALL frames and pixels of a batch element): when the frames / pixels of a batch element
 * are sharded over the ranks of a process group, every rank reduces ITS rows to sums[frames/frames_per_stat][groups][2] = (sum, sum of
 * squares) in double, the ranks add them (RCCL all-reduce of 2 x 32 x batch doubles), and svd_groupnorm_stats_from_sums forms
 * (mean, rstd) with the GLOBAL element count -- svd_groupnorm_apply then runs unchanged on the local rows. */
int svd_groupnorm_sums(const void* X, int64_t ldx, int32_t frames, int32_t pix, int32_t channels, int32_t groups,
                       int32_t frames_per_stat, float* partial, double* sums, int32_t dtype, svd_stream_t stream);
int svd_groupnorm_stats_from_sums(const double* sums, int32_t nstat, int32_t groups, double count, float eps, float* stats,
                                  svd_stream_t stream);

/* LayerNorm over channels per token (eps 1e-5), optional per-frame vector added first (x + vec[frame]) with the
 * sum also written to Xsum (used for "x_mix = x + time_pos_embed" video_attention.py:318-321), optional SiLU
 * (ControlNetConditioningEmbedding per-pixel LN + SiLU, models/control/controlnet.py:108-114).
 * Replaces nn.LayerNorm of attention.py:528-530, video_attention.py:59,87,101-102.
 */
int svd_layernorm(const void* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels,
                  const float* gamma, const float* beta, float eps,
                  const float* addvec, int32_t addvec_ld, int32_t rows_per_vec, void* Xsum, int64_t ldxsum,
                  int32_t silu, int32_t dtype, svd_stream_t stream);

/* ---- layout / elementwise glue ---------------------------------------------------------------------------- */
/* NCHW fp32 (two sources concatenated on C: c0 from X0, c1 from X1, X1 may be NULL) -> channels-last bf16 with
 * channel padding to cpad, scaled per frame by scale[f] (NULL = 1).  (wrappers.py:33 concat + Denoiser c_in) */
int svd_nchw_to_tokens(const float* X0, int32_t c0, const float* X1, int32_t c1, const float* scale,
                       svd_bf16* Y, int32_t cpad, int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream);
/* ---- extended-precision rim (ABI v7, csrc/precision.hip) -------------------------------------------------------
 * SPLIT-3 operands: x_hi = rn16(x), x_lo = rn16(x - x_hi); a row of C values is stored as [hi(C) | lo(C) | hi(C)] and multiplied by
 * weights packed [W_hi | W_hi | W_lo] (per tap for the convolution views): svd_gemm then accumulates A_hi W_hi + A_lo W_hi + A_hi W_lo
 * in fp32 -- ~22-bit operands on the unchanged MFMA kernels.  Used where oracle/ablate_precision_sites.py shows the 16-bit error
 * budget concentrates at negligible FLOPs: ControlNetConditioningEmbedding (models/control/controlnet.py:51-121, once per chunk) and
 * the stem convolutions input_blocks.0.0 (models/diffusion/video_model.py:569, models/control/controlnet.py:524).
 * svd_nchw_to_tokens_x3: svd_nchw_to_tokens with split-3 output rows [frames*pix, 3*cpad]. */
int svd_nchw_to_tokens_x3(const float* X0, int32_t c0, const float* X1, int32_t c1, const float* scale,
                          svd_bf16* Y, int32_t cpad, int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream);
/* fp32 rows [rows, channels <= 512] -> split-3 rows [rows, 3*channels]; flags: per-row LayerNorm over the channels (gamma, beta, eps;
 * the embedding's per-pixel nn.LayerNorm, controlnet.py:108-114) and / or SiLU first, both in fp32. */
enum { SVD_SPLIT3_LAYERNORM = 1, SVD_SPLIT3_SILU = 2 };
int svd_rows_split3(const float* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels, const float* gamma,
                    const float* beta, float eps, int32_t flags, int32_t dtype, svd_stream_t stream);
/* Y = X + B with B fp32 (Merger addition of the fp32 image-condition embedding, controlnet.py:41-42); X / Y 16 bit, or fp32 with
 * dtype | SVD_DTYPE_IN_F32. */
int svd_add_rows_bf32(const void* X, int64_t ldx, const float* B, int64_t ldb, void* Y, int64_t ldy, int64_t rows, int32_t channels,
                      int32_t dtype, svd_stream_t stream);
/* The UNet's head in ONE fp32 kernel: Y[m][0..cout) = conv3x3( SiLU( GroupNorm(X) ) ) + bias, cout <= 4, zero padding
 * (out.0 GroupNorm32 + SiLU + out.2 conv, models/diffusion/video_model.py:493-494,617).  X: channels-last rows of `frames` frames of
 * H x W pixels (16 bit, or fp32 with dtype | SVD_DTYPE_IN_F32); stats: (mean, rstd) per (frame / frames_per_stat, group) as written by
 * svd_groupnorm_stats; Wt: fp32 [9 taps (ky, kx)][channels][4] (cout padded to 4); Y fp32 rows with leading dimension ldy. */
int svd_head_gn_silu_conv3x3(const void* X, int64_t ldx, int32_t frames, int32_t H, int32_t W, int32_t channels, int32_t groups,
                             int32_t frames_per_stat, const float* stats, const float* gamma, const float* beta, const float* Wt,
                             const float* bias, float* Y, int64_t ldy, int32_t cout, int32_t dtype, svd_stream_t stream);
/* channels-last (x_dtype: BF16 | F16 | F32; first c channels of rows with stride ld) -> NCHW fp32 */
int svd_tokens_to_nchw(const void* X, int32_t x_dtype, int64_t ldx, float* Y, int32_t c, int32_t frames,
                       int32_t pix, svd_stream_t stream);
/* Y[m][0..ca) = A[m], Y[m][ca..ca+cb) = B[m]   (th.cat([h, hs.pop()], dim=1), video_model.py:608) */
int svd_concat_channels(const svd_bf16* A, int64_t lda, int32_t ca, const svd_bf16* B, int64_t ldb, int32_t cb,
                        svd_bf16* Y, int64_t ldy, int64_t rows, svd_stream_t stream);
/* Y = X + B (row-wise, same shape)  (Merger addition, controlnet.py:41-42) */
int svd_add_rows(const void* X, int64_t ldx, const svd_bf16* B, int64_t ldb, void* Y, int64_t ldy,
                 int64_t rows, int32_t channels, int32_t dtype, svd_stream_t stream);
/* fp32 rows -> 16-bit rows (channels % 8 == 0): the 16-bit operand copy of an fp32 residual-stream tensor where the reference feeds the
 * stream straight into a convolution / linear (Downsample.op, Upsample.conv, skip_connection, the torch.cat of the decoder half
 * video_model.py:606-611 -- two calls into the two column ranges of Y --, the ControlNet features the CAM mergers project). */
int svd_cast_rows_f32(const float* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int64_t rows, int32_t channels, int32_t dtype,
                      svd_stream_t stream);
/* Y[i_{p0}][i_{p1}][i_{p2}][i_{p3}][:] = X[i0][i1][i2][i3][:] over rows of row_bytes (multiple of 16) bytes: the frame <-> pixel repack
 * around the sequence-parallel all-to-all (replaces the "(b t) ... <-> b ... t" rearranges of video_model.py:75-80 /
 * video_attention.py:289-325 across ranks) in one pass. */
int svd_permute_rows(const void* X, void* Y, int32_t n0, int32_t n1, int32_t n2, int32_t n3, int32_t p0, int32_t p1, int32_t p2,
                     int32_t p3, int64_t row_bytes, svd_stream_t stream);
/* fp32 -> 16-bit cast, optionally through SiLU (emb_layers' leading SiLU, openaimodel.py:284-290) */
int svd_cast_f32(const float* X, svd_bf16* Y, int64_t n, int32_t apply_silu, int32_t dtype, svd_stream_t stream);
/* sinusoidal embedding [cos | sin], freqs = exp(-ln(max_period) * i / half)  (util.py:207-231) -> 16-bit [n][dim], or fp32 with
 * dtype = SVD_DTYPE_F32 (ABI v7: the precision plan's embedding MLPs take it in fp32) */
int svd_timestep_embedding(const float* t, int32_t n, int32_t dim, float max_period, void* Y,
                           int32_t dtype, svd_stream_t stream);

/* ---- sampler glue (Denoiser + LinearPredictionGuider + Euler step) ------------------------------------------
 * One Euler-EDM step tail on the fp32 NCHW state x[T][C][pix]:
 *   den_{u,c} = net_{u,c} * c_out + x * c_skip          (denoiser.py:36-39)
 *   den = den_u + scale[t] * (den_c - den_u)             (guiders.py:78-86)
 *   d = (x - den) / sigma ; x += d * (sigma_next - sigma) (sampling.py:100-103, sampling_utils.py:34-35)
 * net: channels-last fp32 [2T][pix][ldn] (uncond frames first).
 */
int svd_edm_euler_step(float* x, const float* net, int64_t ldn, const float* guidance_scale, int32_t T,
                       int32_t C, int32_t pix, float sigma, float sigma_next, svd_stream_t stream);

/* ---- VAE tail -------------------------------------------------------------------------------------------
 * AE3DConv.time_mix_conv: Conv3d(3->3,(3,1,1)) over frames of the 3-channel image + clamp(-1,1) optional,
 * channels-last fp32 in (ld), NCHW fp32 out.  (temporal_ae.py:99-105, streaming_svd.py:220)
 */
int svd_ae_time_mix3(const float* X, int64_t ldx, const float* w /*[3][3][3] (co,ci,kt)*/, const float* b,
                     float* Y, int32_t frames, int32_t pix, int32_t clamp, svd_stream_t stream);

/* ---- I2VGen-XL enhancement stage ------------------------------------------------------ */
/* nn.AdaptiveAvgPool2d((hout, wout)) on channels-last tokens (image_latents_context_embedding[2],
 * code/i2v_enhance/unet_i2vgen_xl.py:262-269). */
int svd_adaptive_avgpool_tokens(const svd_bf16* X, int64_t ldx, svd_bf16* Y, int64_t ldy, int32_t frames, int32_t hin,
                                int32_t win, int32_t hout, int32_t wout, int32_t channels, int32_t dtype, svd_stream_t stream);
/* I2VGenXLTransformerTemporalEncoder on the 4-channel image latents (unet_i2vgen_xl.py:110-160, call site :700-709):
 * X rows (b, f, p) with >= 4 channels; params = ln_w[4] ln_b[4] wq[8][4] wk[8][4] wv[8][4] wo[4][8] bo[4] w1[16][4] b1[16]
 * w2[4][16] b2[4] (288 floats); Y fp32 NCHW [(b f), 4, pix].  frames <= 128.  X: 16-bit rows, or fp32 rows (ldx in floats, 16-byte aligned)
 * with dtype | SVD_DTYPE_IN_F32 (round 5: the enhancer's precision plan keeps the image-latent projection fp32 up to this kernel). */
int svd_i2v_image_temporal_encoder(const svd_bf16* X, int64_t ldx, const float* params, float* Y, int32_t batch,
                                   int32_t frames, int32_t pix, int32_t dtype, svd_stream_t stream);
/* Classifier-free guidance + one DDIM step (eta 0) on fp32 latents (pipeline_i2vgen_xl.py:872-885 + diffusers
 * DDIMScheduler.step): v = pu + g (pc - pu) (pc may be NULL: no guidance); out = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps. */
int svd_ddim_cfg_step(const float* x, const float* pred_uncond, const float* pred_cond, float* out, int64_t n,
                      float guidance_scale, float alpha_t, float alpha_prev, int32_t v_prediction, svd_stream_t stream);

/* Row A13: fp32 frames [frames][3][pix] in [-1, 1] -> uint8 [frames][pix][3], the reference's exact sequence of fp32 roundings:
 * convert_range (utils/result_processor.py:4-14, diffusion_trainer/streaming_svd.py:353) + IImage/torch2np truncation
 * (lib/farancia/libimage/iimage.py:35-36).  Bit-exact with the reference. */
int svd_frames_to_uint8(const float* X, uint8_t* Y, int32_t frames, int32_t pix, svd_stream_t stream);

/* Fused GEGLU feed-forward of the 320-channel level (ABI v8, round 5):
 *     Y = [alpha * S + (1 - alpha) *] ( R + b2 + W2 . ( (W1v X + b1v) * gelu_erf(W1g X + b1g) ) )
 * Replaces the two svd_gemm launches of FeedForward(dim 320, mult 4, glu) -- GEGLU.proj + gate and net[2] (+ the residual add of the
 * transformer block, + the AlphaBlender of the temporal block): models/svd/sgm/modules/attention.py:94-120,567-593,
 * models/svd/sgm/modules/video_attention.py:125-168; i2v_enhance/attention.py:414-534 (diffusers FeedForward, activation_fn "geglu").
 * The [M, hidden] activation never reaches HBM (csrc/ff_fused.hip).
 *   X  [M][ldx] 16-bit rows (the LayerNorm output), channels == 320.
 *   Wp packed image of W1 (2*hidden x 320: value rows, then gate rows), b1 (2*hidden) and W2 (320 x hidden):
 *      svd_ff_fused_pack_bytes(hidden) bytes = hidden/32 chunks of [40 W1 fragments | 64 biases (1 KiB slot) | 20 W2 fragments] in MFMA
 *      fragment order (streamingt2v_amd/video_model.pack_ff_fused writes it; layout documented in csrc/ff_fused.hip).  hidden % 64 == 0.
 *   b2 [320] fp32.  R: residual rows or NULL; S: blend partner rows or NULL (alpha ignored then); both fp32 when res_f32 else 16 bit.
 *   Y  fp32 rows when out_f32, else 16-bit rows.  The hidden activation is rounded to the 16-bit type exactly like the two-launch path.
 *   rowvec (ABI v9): optional per-frame vector rows (fp32, leading dimension rowvec_ld >= 320) added to the result, row m taking vector m / rows_per_vec
 *      (rows_per_vec % 32 == 0): `x + time_pos_embed` of the time_stack enters as R + rowvec, the sum is never materialised (video_attention.py:318-321).
 *   Yn (ABI v10): optional second output (16-bit rows, leading dimension ldyn >= 320) = LayerNorm(Y + ln_addvec[m / ln_rows_per_vec]) * ln_gamma + ln_beta,
 *      the nn.LayerNorm that consumes the block's result (norm_in of the time_stack over x + time_pos_embed behind the spatial ff, norm1 behind ff_in:
 *      video_attention.py:125-168,318-321); only with an fp32 residual, fp32 output and no blend (SVD_EINVAL otherwise).  ln_addvec may be NULL. */
int64_t svd_ff_fused_pack_bytes(int32_t hidden);
int svd_ff_geglu_fused(const svd_bf16* X, int64_t ldx, const void* Wp, int32_t channels, int32_t hidden, const float* b2,
                       const void* R, int64_t ldr, const void* S, int64_t lds, float alpha, int32_t res_f32,
                       void* Y, int64_t ldy, int32_t out_f32, int64_t M, int32_t dtype,
                       const float* rowvec, int32_t rowvec_ld, int32_t rows_per_vec,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, const float* ln_addvec, int32_t ln_addvec_ld, int32_t ln_rows_per_vec,
                       svd_bf16* Yn, int64_t ldyn, svd_stream_t stream);

/* Row-resident 320 -> N projection (ABI v11, round 6, csrc/rowproj.hip):  Y[M, N] = X[M, 320] . W^T + bias, 16-bit rows in and out, N % 64 == 0.
 * Replaces svd_gemm for the q | k (N = 640) and q | k | v (N = 960) projections of the 320-channel transformer blocks (attn1.to_q / to_k of
 * SpatialVideoTransformer, to_q / to_k / to_v of its time_stack: sgm/modules/attention.py:246-262, video_attention.py:125-168; the enhancer's Transformer2DModel /
 * TransformerTemporalModel): a wave keeps its 32 token rows in registers for the whole width, so X is read once instead of once per N tile.
 *   Wp   packed weights (svd_rowproj320_pack_bytes(N) bytes, video_model.pack_rowproj320): chunk ch of 64 channels = 40 fragments of 1 KiB, fragment 2 s + t
 *        (k-step s of 20, tile t of 2), lane l: W[64 ch + 2 (l % 32) + t][16 s + 8 (l / 32) .. + 7] in the element type
 *   bias fp32 [N] or NULL;  ldx % 8 == 0, ldy % 2 == 0 (elements), Y 4-byte aligned */
int64_t svd_rowproj320_pack_bytes(int32_t N);
int svd_rowproj320(const svd_bf16* X, int64_t ldx, const void* Wp, const float* bias, svd_bf16* Y, int64_t ldy, int64_t M, int32_t N, int32_t dtype,
                   svd_stream_t stream);

/* Row-owning 320 -> 320 projection of the fp32 residual stream, optionally with the LayerNorm that follows it (ABI v9, round 6):
 *     V  = R + bias + rowvec[row / rows_per_vec] + X . W^T                  -> Y  (fp32 rows when out_f32, else 16-bit rows; may be NULL when Yn is given)
 *     Yn = LayerNorm(V; ln_gamma, ln_beta, ln_eps)                          -> 16-bit rows (NULL: no LayerNorm)
 * Replaces svd_gemm (+ the svd_layernorm behind it) for proj_in / attn1.to_out / proj_out of SpatialVideoTransformer and the attn1.to_out of its
 * time_stack at dim 320: models/svd/sgm/modules/video_attention.py:260-333,125-168, attention.py:567-593 (norm1 / norm3: attention.py:528-530); the same
 * projections of the enhancer's Transformer2DModel / TransformerTemporalModel (i2v_enhance/transformer_temporal.py:121-200).  csrc/rowgemm.hip: a wave owns
 * 32 token rows and all 320 outputs, so the normalisation is lane-local and the fp32 tensor is not re-read.
 *   X  [M][ldx] 16-bit rows, 320 channels.   Wp: svd_rowgemm320_pack_bytes() bytes, 200 MFMA fragments of W [320 out][320 in] (fragment 10 s + o: lane l holds
 *      W[32 o + l % 32][16 s + 8 (l / 32) .. + 7]; streamingt2v_amd/video_model.pack_rowgemm320).   bias [320] fp32 or NULL.
 *   rowvec: per-frame vector rows (fp32, leading dimension rowvec_ld) or NULL; rows_per_vec % 32 == 0.   R: fp32 residual rows or NULL. */
int64_t svd_rowgemm320_pack_bytes(void);
int svd_rowgemm320(const svd_bf16* X, int64_t ldx, const void* Wp, const float* bias, const float* rowvec, int32_t rowvec_ld, int32_t rows_per_vec,
                   const float* R, int64_t ldr, void* Y, int64_t ldy, int32_t out_f32, const float* ln_gamma, const float* ln_beta, float ln_eps,
                   svd_bf16* Yn, int64_t ldyn, int64_t M, int32_t dtype, svd_stream_t stream);

/* Exact-erf GELU in place on 16-bit rows (nn.GELU of the OpenCLIP ViT-H/14 MLP inside FrozenOpenCLIPImageEmbedder,
 * models/svd/sgm/modules/encoders/modules.py:574-732 -> open_clip transformer.py ResidualAttentionBlock.mlp). */
int svd_gelu_rows(svd_bf16* X, int64_t ldx, int64_t rows, int32_t channels, int32_t dtype, svd_stream_t stream);

/* ---- EMA-VFI frame interpolation (code/i2v_enhance/thirdparty/VFI, called from inference_i2v.py:211-224 through
 *      i2v_enhance_interface.vfi_process :30-61).  Convolutions, linears and LayerNorms of the model run on svd_gemm / svd_layernorm;
 *      the entries below are the remaining operators.  "rows" tensors are channels-last 16-bit [pixels][ld]. ---- */

/* nn.PReLU(channels) in place (model/refine.py:8-19, model/flow_estimation.py:9-14, model/feature_extractor.py:293-305); dtype: a 16-bit
 * element type (channels % 8 == 0) or SVD_DTYPE_F32. */
int svd_prelu_rows(void* X, int64_t ldx, int64_t rows, int32_t channels, const float* slope, int32_t dtype, svd_stream_t stream);

/* Mlp.dwconv + act: depthwise 3x3 (padding 1, bias) followed by exact GELU (model/feature_extractor.py:104-108, 505-515).
 * w9: [9][channels] fp32 (tap-major: w9[(ky*3+kx)*channels + c] = weight[c][0][ky][kx]). */
int svd_dwconv3x3_gelu(const svd_bf16* X, int64_t ldx, svd_bf16* Y, int64_t ldy, const float* w9, const float* bias, int32_t frames,
                       int32_t h, int32_t w, int32_t channels, int32_t dtype, svd_stream_t stream);

/* InterFrameAttention.forward on 7x7 windows, head dim 32 (model/feature_extractor.py:141-171).  Q [n_win*49][ldq], KV [n_win*49][ldkv]
 * (k | v, each heads*32 wide), CE [n_win*49][ldce] fp32 = cor_embed (heads * motion_per_head wide, motion_per_head <= 16).  Window w
 * attends to the keys/values of window (w + n_win/2) % n_win (the other frame, :264).  mask: [n_mask][49][49] fp32 or NULL; window w
 * uses mask[w % n_mask].  OX = softmax(q k^T scale + mask) v;  OC = softmax(..) ce - ce  (c_reverse - cor_embed_, :166-168). */
int svd_window_attn_7x7(const svd_bf16* Q, int64_t ldq, const svd_bf16* KV, int64_t ldkv, const float* CE, int64_t ldce, const float* mask,
                        int32_t n_mask, svd_bf16* OX, int64_t ldo, svd_bf16* OC, int64_t ldc, int32_t n_win, int32_t heads,
                        int32_t motion_per_head, float scale, int32_t dtype, svd_stream_t stream);

/* warp(tenInput, tenFlow) (model/warplayer.py:7-22): grid_sample(bilinear, border, align_corners=True) at (x + flow[0], y + flow[1]).
 * X/Y channels-last [frames*h*w][ld]; dtype SVD_DTYPE_F32 (any channel count) or a 16-bit type (channels % 8 == 0);
 * flow: fp32, two values per pixel at flow[pixel * ldf]. */
int svd_warp_bilinear(const void* X, int64_t ldx, void* Y, int64_t ldy, const float* flow, int64_t ldf, int32_t frames, int32_t h, int32_t w,
                      int32_t channels, int32_t dtype, svd_stream_t stream);

/* F.interpolate(scale_factor, mode="bilinear", align_corners=False) on channels-last fp32 (model/flow_estimation.py:30-39,64):
 * Y[c] = (accumulate ? Y[c] : 0) + mult[c] * interp(X)[c]  (mult NULL = 1); hout/wout = floor(in * scale_factor) is the caller's. */
int svd_resize_bilinear_f32(const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t frames, int32_t hin, int32_t win, int32_t hout,
                            int32_t wout, int32_t channels, float scale_factor, const float* mult, int32_t accumulate, svd_stream_t stream);

/* merged = w0 sigmoid(mask) + w1 (1 - sigmoid(mask)); pred = clamp(merged + sigmoid(unet_out[:3]) * 2 - 1, 0, 1)
 * (model/flow_estimation.py:131-139, model/refine.py:71).  warped0/1, merged (may be NULL), pred: fp32 [n_pixels][3]. */
int svd_vfi_merge(const float* warped0, const float* warped1, const float* mask, int64_t ld_mask, const float* unet_out, int64_t ld_unet,
                  float* merged, float* pred, int64_t n_pixels, svd_stream_t stream);

/* fast-TTA average (Trainer.py:90-94): out = (pred2[0] + rot180(pred2[1])) / 2 on [2][h][w][3] fp32; out_u8 (optional) =
 * (uint8)(out * 255.0f), the truncation of i2v_enhance_interface.py:46-47. */
int svd_vfi_tta_average(const float* pred2, float* out, uint8_t* out_u8, int32_t h, int32_t w, svd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SVDHIP_H */
