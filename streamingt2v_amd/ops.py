"""Tensor-level launchers for the libsvdhip.so kernels.

PyTorch is plumbing only here: it owns device memory and the stream.  Every function below passes raw
pointers + sizes + the current HIP stream through the C ABI (include/svdhip.h); no arithmetic happens in torch.
Activations are "token" tensors: shape [rows, C] (rows = frames * pixels), bf16, channels contiguous.
"""
import ctypes as C

import torch

from . import lib as _l
from .lib import A_CONV3X3, A_PLAIN, A_TEMPORAL3, EPI_GEGLU, EPI_SILU, OUT_BF16, OUT_BF16_T, OUT_F32, GemmArgs, check

_lib = _l.lib
BF16 = torch.bfloat16
F16 = torch.float16
_DT = {torch.bfloat16: _l.DTYPE_BF16, torch.float16: _l.DTYPE_F16, torch.float32: _l.DTYPE_F32}

# Element type new 16-bit tensors are created in when no 16-bit input exists to inherit it from (casts of fp32 inputs,
# packed weights).  fp16 is the default: it is the reference's own autocast type (config.yaml:8 "16-mixed"), the type every parity
# tolerance of tests/test_gpu_*parity*.py is asserted in and the type bench.py / __graft_entry__.smoke() run in (same MFMA rate as bf16,
# 8x finer rounding).  set_element_dtype(torch.bfloat16) selects bf16 (~8x the fp16 deviation from the fp32 reference).
DEFAULT_ELEM = torch.float16
ELEM = DEFAULT_ELEM


# fp32 RESIDUAL STREAM (round 3, optional): the tensors the reference's residual additions run on (ResBlock h + skip, transformer x + attn(x),
# x + ff(x), the alpha-blends; openaimodel.py:351-354, attention.py:567-593, video_attention.py:125-168) stay fp32 BETWEEN the kernels of
# the VideoUNet / ControlNet: GEMM epilogues read the residual and write the sum in fp32 (svd_gemm_args.res_f32 / SVD_OUT_F32: the separate
# stream instantiation of the GEMM kernel), the norms read fp32 (SVD_DTYPE_IN_F32) and write the 16-bit GEMM operand.  Only what a matrix
# core consumes is rounded to 16 bit -- the "operand-only floor" of oracle/measure_precision_floor.py.
# Measured at the shipped size (profiles/r03_parity_report.txt, r03_stream_cost.txt): StreamingWrapper.forward vs the reference's fp32 output
# 0.86e-3 mean / 1.07e-3 max per-frame L2 with the fp32 stream against 1.15e-3 / 1.39e-3 with the 16-bit stream -- and against 1.42e-3 / 1.68e-3
# for the REFERENCE'S OWN fp16-autocast execution (tests/golden/wrapper_fullsize_autocast.json).  The stream's bytes double (norm reads,
# residual reads, output writes) and the job runs ~10 % slower, far above the 3 % the round-2 review allowed for making it the default:
# it is an option (set_stream_f32(True), bench.py --residual-stream fp32); the default is the 16-bit stream of round 2, whose deviation is
# inside the reference's own production-precision envelope.
STREAM_F32 = False


def set_stream_f32(on=True):
    global STREAM_F32
    STREAM_F32 = bool(on)


# PRECISION PLAN (round 4).  oracle/ablate_precision_sites.py emulates the 16-bit execution of StreamingWrapper.forward on the CPU oracle and
# keeps one site of the network exact at a time: the deviation from the reference's fp32 path concentrates in places that cost next to nothing --
#   EXACT_RIM   the ControlNet's image-condition embedding (15 % of the squared error; computed once per chunk), the two stem convolutions and
#               the UNet's head (7.5 %; GroupNorm + SiLU + conv to 4 channels) run with split-3 (~22-bit) MFMA operands / in fp32 (csrc/precision.hip);
#   CN_STREAM_F32  the fp32 residual stream INSIDE THE CONTROLNET only (16 % of the squared error for ~1/8 of the stream's bytes: the ControlNet
#               sees 14 of the 64 frames of a forward and only the encoder half).
#   STREAM_F32_MIN_CH  the fp32 residual stream in the UNet blocks with at least this many channels (0 = none).  The DEFAULT, 320, equals
#               model_channels: EVERY block and convolution of the UNet carries the stream in fp32 -- the same as STREAM_F32 = True for the UNet --
#               at +6.9 % of a forward, because nothing cheaper keeps the worst frame of A5 under 1e-3 (profiles/r04_fullsize_parity_plans.txt:
#               640 -> 0.940e-3 mean / 1.116e-3 max at +3.6 %; 1280 -> rim + ControlNet stream only, 1.018e-3 / 1.201e-3 at +1.1 %).  The stream's
#               cost is its bytes: a level-0 tensor (320 channels @ 72x128) is 295 MB, level 1 147 MB, levels 2 / 3 (1280 channels) 74 / 18 MB.
# All are package defaults (environment overrides for A/B runs: SVD_EXACT_RIM, SVD_CN_STREAM_F32, SVD_STREAM_F32_MIN_CH); STREAM_F32 forces the
# stream on in every network evaluated (it is what stream_scope sets inside the ControlNet).
# WHEN THEY TAKE EFFECT: EXACT_RIM at load_state_dict (the rim packs split-3 weights); CN_STREAM_F32 / STREAM_F32_MIN_CH / STREAM_F32 are read per forward.
import os as _os
EXACT_RIM = _os.environ.get("SVD_EXACT_RIM", "1") != "0"
CN_STREAM_F32 = _os.environ.get("SVD_CN_STREAM_F32", "1") != "0"
STREAM_F32_MIN_CH = int(_os.environ.get("SVD_STREAM_F32_MIN_CH", "320"))
# per block kind (A/B sweeps only; None = STREAM_F32_MIN_CH): the ResBlocks' / the transformers' own threshold
STREAM_F32_MIN_CH_KIND = {"res": None, "svt": None}
for _k in ("res", "svt"):
    if _os.environ.get("SVD_STREAM_F32_MIN_CH_" + _k.upper()):
        STREAM_F32_MIN_CH_KIND[_k] = int(_os.environ["SVD_STREAM_F32_MIN_CH_" + _k.upper()])


def set_precision_plan(exact_rim=None, cn_stream_f32=None, stream_f32_min_ch=None):
    """Select the round-4 precision plan (None = leave).  exact_rim takes effect at load_state_dict (the rim packs its own weights: call this
    BEFORE loading, like set_element_dtype); cn_stream_f32 and stream_f32_min_ch are read at every forward."""
    global EXACT_RIM, CN_STREAM_F32, STREAM_F32_MIN_CH
    if exact_rim is not None:
        EXACT_RIM = bool(exact_rim)
    if cn_stream_f32 is not None:
        CN_STREAM_F32 = bool(cn_stream_f32)
    if stream_f32_min_ch is not None:
        STREAM_F32_MIN_CH = int(stream_f32_min_ch)


# transformers below their own threshold but at / above this one keep only their BLOCK OUTPUT (x + proj_out(...), the tensor the next block's residual
# chain starts from) in fp32: one extra 4-byte tensor per block instead of seven (A/B sweeps; 0 = off)
STREAM_F32_SVT_IO_MIN_CH = int(_os.environ.get("SVD_STREAM_F32_SVT_IO_MIN_CH", "0"))


def stream_on(channels, kind=None):
    """Does a block of `channels` channels keep its residual stream in fp32?  (STREAM_F32: every block; else the precision plan's threshold.)"""
    if STREAM_F32:
        return True
    m = STREAM_F32_MIN_CH_KIND.get(kind)
    m = STREAM_F32_MIN_CH if m is None else m
    return m > 0 and channels >= m


# ENHANCER PRECISION PLAN (round 5): the same two instruments inside I2VGenXLUNet (i2vgen_unet.py; unet_i2vgen_xl.py:573-814).
#   I2V_EXACT_RIM          conv_in (8 -> 320 channels), the time / fps embedding MLPs and the three image_latents_proj_in convolutions with split-3
#                          operands; conv_norm_out + SiLU + conv_out (320 -> 4 channels) as the one fp32 head kernel.  Read at load_state_dict time.
#   I2V_STREAM_F32_MIN_CH  the fp32 residual stream in every block with at least this many channels (0 = none): ResnetBlock2D output, TemporalConvLayer
#                          identity add, the transformers' x + attn(x) / x + ff(x) chains and their block outputs, the down / up-sampler outputs.
#                          Read per forward.
# Environment overrides for A/B runs: SVD_I2V_EXACT_RIM, SVD_I2V_STREAM_F32_MIN_CH.
I2V_EXACT_RIM = _os.environ.get("SVD_I2V_EXACT_RIM", "1") != "0"
I2V_STREAM_F32_MIN_CH = int(_os.environ.get("SVD_I2V_STREAM_F32_MIN_CH", "320"))


def set_i2v_precision_plan(exact_rim=None, stream_f32_min_ch=None):
    """The enhancer's precision plan (None = leave).  exact_rim takes effect at I2VGenXLUNet.load_state_dict (it packs its own weights),
    stream_f32_min_ch at the next forward."""
    global I2V_EXACT_RIM, I2V_STREAM_F32_MIN_CH
    if exact_rim is not None:
        I2V_EXACT_RIM = bool(exact_rim)
    if stream_f32_min_ch is not None:
        I2V_STREAM_F32_MIN_CH = int(stream_f32_min_ch)


def i2v_stream_on(channels):
    return I2V_STREAM_F32_MIN_CH > 0 and channels >= I2V_STREAM_F32_MIN_CH


# DECODER PRECISION PLAN (round 6): the same two instruments inside the temporal VideoDecoder (temporal_ae.py; code/models/svd/sgm/modules/autoencoding/
# temporal_ae.py:291-347).  The reference decodes in fp32 (`disable_first_stage_autocast: true`, config.yaml:310): against the fp16-autocast envelope of a whole
# chunk the decoder's own 16-bit deviation (0.86e-3 per frame) is what put the DECODED frames' mean above the envelope's while the latents were 22 % inside it
# (profiles/r06_fullsize_chunk_tests.txt).
#   AE_EXACT_RIM          conv_in (4 -> 512 channels) with split-3 operands, norm_out + SiLU + conv_out (128 -> 3 channels) as the one fp32 head kernel.
#                         Read at load_state_dict time (together with EXACT_RIM, which packs the split-3 weights).
#   AE_STREAM_F32_MIN_CH  the fp32 residual stream (ResnetBlock h + skip, the time_stack's alpha blend, AttnBlock x + proj_out) in the blocks with at least this
#                         many output channels (0 = none).  Read per forward.
# Environment overrides for A/B runs: SVD_AE_EXACT_RIM, SVD_AE_STREAM_F32_MIN_CH.
AE_EXACT_RIM = _os.environ.get("SVD_AE_EXACT_RIM", "1") != "0"
AE_STREAM_F32_MIN_CH = int(_os.environ.get("SVD_AE_STREAM_F32_MIN_CH", "128"))


def set_ae_precision_plan(exact_rim=None, stream_f32_min_ch=None):
    global AE_EXACT_RIM, AE_STREAM_F32_MIN_CH
    if exact_rim is not None:
        AE_EXACT_RIM = bool(exact_rim)
    if stream_f32_min_ch is not None:
        AE_STREAM_F32_MIN_CH = int(stream_f32_min_ch)


def ae_stream_on(channels):
    return AE_STREAM_F32_MIN_CH > 0 and channels >= AE_STREAM_F32_MIN_CH


class stream_scope:
    """with stream_scope(True): the fp32 residual stream is on for the networks evaluated inside (ControlNet.forward_tokens)."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global STREAM_F32
        self.prev = STREAM_F32
        STREAM_F32 = self.prev or self.on

    def __exit__(self, *exc):
        global STREAM_F32
        STREAM_F32 = self.prev
        return False


def set_element_dtype(dt=None):
    """Select the 16-bit element type of everything created from here on (None: back to DEFAULT_ELEM).  Call it BEFORE load_state_dict:
    packed weights keep the type they were packed in."""
    global ELEM
    dt = DEFAULT_ELEM if dt is None else dt
    assert dt in (torch.bfloat16, torch.float16)
    ELEM = dt


def _dt(t):
    return _DT[t.dtype]


def _dt_in(x):
    """dtype argument of a norm / add_rows whose input may be the fp32 residual stream: 16-bit element type (| IN_F32)."""
    return (_DT[ELEM] | _l.DTYPE_IN_F32) if x.dtype == torch.float32 else _DT[x.dtype]


def _odt(x):
    """16-bit type of the output a norm derives from x."""
    return ELEM if x.dtype == torch.float32 else x.dtype


def tensor_version(t):
    """In-place version counter of `t` for the per-chunk caches (video_model / wrappers key them on tensor IDENTITY + version), or None
    when the tensor has none (tensors created under torch.inference_mode raise on ._version): None means "never a cache hit".
    Contract of those caches: the same tensor OBJECT with the same version holds the same values -- writes through
    .data / set_() do not bump the version and are not seen; call reset_caches() on the networks after such a write or between videos."""
    try:
        return t._version
    except RuntimeError:
        return None

_zeros = {}
_gn_ws = {}
# Work log (tools/roofline_table.py): SVD_WORKLOG=<path> makes every launcher below add its algorithmic FLOP / HBM bytes to a per-kernel
# total for the WHOLE process and dump {kernel key: [launches, flop, bytes]} as JSON at exit.  Run under `rocprofv3 --kernel-trace` the
# same process gives the device time per kernel name: time and work cover exactly the same dispatches, so TFLOP/s, TB/s and the fraction of
# the roofline of every kernel can be recomputed from profiles/ alone.  None (default) = no bookkeeping.
worklog = None
if __import__("os").environ.get("SVD_WORKLOG"):
    worklog = {}

    def _dump_worklog(path=__import__("os").environ["SVD_WORKLOG"]):
        import json
        with open(path, "w") as f:
            json.dump(worklog, f)
    __import__("atexit").register(_dump_worklog)


def _wl(key, flops=0.0, nbytes=0.0):
    e = worklog.get(key)
    if e is None:
        e = worklog[key] = [0, 0.0, 0.0]
    e[0] += 1; e[1] += flops; e[2] += nbytes


trace = None   # bench.py installs a per-launch HIP-event recorder here (None = zero overhead)
tuner = None   # tools/tune_gemm.py installs an in-situ tile autotuner here


def _load_tile_table():
    """Measured best tile config per GEMM signature (written by tools/tune_gemm.py on an MI355X).  Signatures that
    are not in the table use the heuristic of svd_gemm_pick_config."""
    import json
    import os
    # SVD_GEMM_TILES=<file name next to this module> selects another table (A/B of a re-tuned table against the committed one)
    name = os.path.basename(os.environ.get("SVD_GEMM_TILES", "gemm_tiles.json"))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {k: int(v["cfg"]) for k, v in json.load(f)["table"].items()}


def gemm_signature(a):
    # the tile choice does not depend on the element type (bf16 and fp16 MFMA have the same shape and rate)
    return f"m{a.a_mode}_M{a.M}_N{a.N}_K{a.K}_s{a.stride}_u{a.ups}_e{a.epi_flags}_o{a.out_mode}"      # (res_f32 rides with o1: same tile ranking)


_tile_table = _load_tile_table()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def zeros_page(device):
    z = _zeros.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.int16, device=device)   # all-zero bits are 0.0 in bf16 and fp16
        _zeros[device] = z
    return z


def _rows_ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "token tensor must be [rows, C] with contiguous channels"
    return t.shape[0], t.stride(0)


def gemm(a, w, *, bias=None, rowvec=None, rows_per_vec=0, residual=None, blend=None, geglu=False, silu=False,
         out=None, out_f32=False, conv=None, temporal=None, trans_out=None, n_out=None, tile_cfg=0, k=None):
    """C = epilogue(A_view . W^T).  See svd_gemm in include/svdhip.h.

    a: [rows_src, lda] bf16 token tensor (for conv/temporal views: the SOURCE tensor).
    w: [N, K] bf16.   conv = dict(cin, hin, win, hout, wout, stride, ups, frames)   temporal = dict(cin, T, pix, M)
    blend = (alpha: float, S tensor).   trans_out = dict(tok_per_frame, tokens_ld, out=tensor [frames, N, tokens_ld])
    """
    assert a.dtype in (BF16, F16) and w.dtype == a.dtype and a.is_cuda and w.is_cuda
    edt = a.dtype
    N, K = w.shape[0], (k if k is not None else w.shape[1])
    args = GemmArgs()
    args.A, args.lda = a.data_ptr(), a.stride(0)
    args.W, args.ldw = w.data_ptr(), w.stride(0)
    args.zeros = zeros_page(a.device).data_ptr()
    if conv is not None:
        M = conv["frames"] * conv["hout"] * conv["wout"]
        args.a_mode = A_CONV3X3
        args.cin, args.hin, args.win = conv["cin"], conv["hin"], conv["win"]
        args.hout, args.wout = conv["hout"], conv["wout"]
        args.stride, args.ups = conv.get("stride", 1), conv.get("ups", 0)
        args.pad_mode = conv.get("pad_mode", 0)
        assert a.shape[0] == conv["frames"] * conv["hin"] * conv["win"]
    elif temporal is not None:
        M = a.shape[0]
        args.a_mode = A_TEMPORAL3
        args.cin, args.t_frames, args.rows_per_frame = temporal["cin"], temporal["T"], temporal["pix"]
    else:
        M = a.shape[0]
        args.a_mode = A_PLAIN
    args.M, args.N, args.K = M, N, K
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
        args.bias = bias.data_ptr()
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.stride(-1) == 1
        args.rowvec, args.rowvec_ld, args.rows_per_vec = rowvec.data_ptr(), rowvec.stride(0), rows_per_vec
    r32 = None
    if residual is not None:
        assert residual.dtype in (edt, torch.float32) and residual.stride(1) == 1
        r32 = residual.dtype == torch.float32
        args.R, args.ldr = residual.data_ptr(), residual.stride(0)
    if blend is not None:
        alpha, S = blend
        assert S.dtype in (edt, torch.float32) and S.stride(1) == 1
        assert r32 is None or r32 == (S.dtype == torch.float32), "residual and blend partner must both be 16-bit or both fp32 (the stream)"
        r32 = S.dtype == torch.float32
        args.S, args.lds, args.alpha = S.data_ptr(), S.stride(0), float(alpha)
    args.res_f32 = int(bool(r32))
    nout = N // 2 if geglu else N
    if n_out is not None:
        nout = n_out
    if geglu:
        args.epi_flags = EPI_GEGLU
    if silu:
        args.epi_flags |= EPI_SILU
    if trans_out is not None:
        o = trans_out["out"]
        assert o.dtype == edt
        args.C, args.ldc = o.data_ptr(), 0
        args.out_mode = OUT_BF16_T
        args.tok_per_frame, args.tokens_ld = trans_out["tok_per_frame"], trans_out["tokens_ld"]
        out = o
    else:
        if out is None:
            out = torch.empty((M, nout), dtype=torch.float32 if out_f32 else edt, device=a.device)
        assert out.shape[0] == M and out.stride(1) == 1
        args.C, args.ldc = out.data_ptr(), out.stride(0)
        args.out_mode = OUT_F32 if out.dtype == torch.float32 else OUT_BF16
        assert out.dtype in (torch.float32, edt)
    args.dtype = _DT[edt]
    if tile_cfg == 0:
        if tuner is not None:
            tile_cfg = tuner.select(args, _stream())
        elif _tile_table:
            sig = gemm_signature(args)
            tile_cfg = _tile_table.get(sig, 0)
            if tile_cfg == 0 and sig.endswith("_o1"):       # fp32-stream producer not tuned yet: the ranking of its 16-bit-output twin
                tile_cfg = _tile_table.get(sig[:-1] + "0", 0)
    args.tile_cfg = tile_cfg
    if trace is not None or worklog is not None:
        cfg = tile_cfg or _lib.svd_gemm_pick_config(C.byref(args))
        # algorithmic HBM bytes of this launch: the source activation and the weights read once, the output (and the residual /
        # blend operands) moved once
        esz = 2
        src_rows = a.shape[0] if conv is None else conv["frames"] * conv["hin"] * conv["win"]
        src_cols = K if (conv is None and temporal is None) else args.cin
        nbytes = (src_rows * src_cols + N * K) * esz + M * nout * (4 if out.dtype == torch.float32 else esz)
        nbytes += M * nout * (4 if r32 else esz) * ((residual is not None) + (blend is not None))
        if worklog is not None:
            _wl(f"gemm|{cfg}|{3 if (args.a_mode == A_CONV3X3 and args.ups) else args.a_mode}|{'f16' if edt == F16 else 'bf16'}", 2.0 * M * N * K, float(nbytes))
    if trace is not None:
        with trace.launch(f"gemm_cfg{cfg}_mode{args.a_mode}", flops=2.0 * M * N * K, sig=gemm_signature(args), nbytes=float(nbytes)):
            check(_lib.svd_gemm(C.byref(args), _stream()), f"svd_gemm(M={M},N={N},K={K},mode={args.a_mode})")
        return out
    check(_lib.svd_gemm(C.byref(args), _stream()), f"svd_gemm(M={M},N={N},K={K},mode={args.a_mode})")
    return out


# FUSED FEED-FORWARD (round 5, csrc/ff_fused.hip): LayerNorm output -> GEGLU projection -> gate -> down-projection (+ residual, + blend) in ONE
# launch for the 320-channel blocks; the [M, 1280] hidden activation never reaches HBM.  SVD_FF_FUSED=0 keeps the two svd_gemm launches (A/B).
FF_FUSED = _os.environ.get("SVD_FF_FUSED", "1") != "0"
# the LayerNorm behind a feed-forward from the fused kernel's own epilogue (round 6, ABI v10).  Built, parity-tested -- and OFF: 1.54 ms against 1.27 + 0.19 ms for the
# two kernels at M = 460 800, -0.5 % on the stage-1 job (profiles/r06_ff_ln_kernel_ab.txt, r06_bench6_ff_ln_ab.txt): the kernel's eight waves reach the epilogue together,
# the second store stream and the 22 spilled registers are exposed there.  SVD_FF_FUSED_LN=1 selects it.
FF_FUSED_LN = _os.environ.get("SVD_FF_FUSED_LN", "0") == "1"


def ff_fused_ok(channels, hidden):
    return FF_FUSED and channels == 320 and hidden % 64 == 0


def ff_geglu_fused(x, img, hidden, b2, *, residual=None, blend=None, out_f32=False, out=None, rowvec=None, rows_per_vec=0, ln=None, ln_eps=1e-5, ln_addvec=None,
                   ln_rows_per_vec=0):
    """x [M, 320] 16-bit rows; img = video_model.pack_ff_fused(...) on the device (uint8); b2 [320] fp32; residual [M, 320] (16 bit or fp32);
    blend = (alpha, S) like ops.gemm.  Returns residual + b2 + W2 (value * gelu(gate)) [blended], fp32 when out_f32 else 16 bit.
    ln = (gamma, beta) fp32 [320]: also returns LayerNorm(result + ln_addvec[row // ln_rows_per_vec]) in the 16-bit type -> (y, yn); fp32 residual and output only."""
    assert x.dtype in (BF16, F16) and x.is_cuda and x.stride(1) == 1 and x.shape[1] == 320 and img.dtype == torch.uint8
    M, Cc = x.shape
    r32 = None
    R = S = None
    alpha = 0.0
    if residual is not None:
        assert residual.dtype in (x.dtype, torch.float32) and residual.stride(1) == 1 and residual.shape == x.shape
        r32, R = residual.dtype == torch.float32, residual
    if blend is not None:
        alpha, S = blend
        assert R is not None and S.dtype == R.dtype and S.stride(1) == 1 and S.shape == x.shape
    if out is None:
        out = torch.empty((M, Cc), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    assert out.shape == (M, Cc) and out.stride(1) == 1 and out.dtype in (torch.float32, x.dtype)
    flops = 2.0 * M * (2 * hidden) * Cc + 2.0 * M * hidden * Cc
    nbytes = float(M * Cc * (2 + out.element_size() + (R.element_size() if R is not None else 0) + (S.element_size() if S is not None else 0) + (2 if ln is not None else 0))
                   + img.numel())
    if worklog is not None:
        _wl("ff_geglu_fused_kernel", flops, nbytes)
    yn = g = b = None
    if ln is not None:
        g, b = ln
        assert r32 and out.dtype == torch.float32 and S is None and g.dtype == b.dtype == torch.float32 and g.numel() == b.numel() == Cc
        yn = torch.empty((M, Cc), dtype=x.dtype, device=x.device)
        if ln_addvec is not None:
            assert ln_addvec.dtype == torch.float32 and ln_addvec.stride(-1) == 1 and ln_rows_per_vec > 0 and ln_rows_per_vec % 32 == 0
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.stride(-1) == 1 and rows_per_vec > 0 and rows_per_vec % 32 == 0
    args = (_p(x), x.stride(0), _p(img), Cc, hidden, _p(b2), _p(R), R.stride(0) if R is not None else 0, _p(S), S.stride(0) if S is not None else 0,
            float(alpha), int(bool(r32)), _p(out), out.stride(0), int(out.dtype == torch.float32), M, _dt(x),
            _p(rowvec), rowvec.stride(0) if rowvec is not None else 0, rows_per_vec,
            _p(g), _p(b), float(ln_eps), _p(ln_addvec if ln is not None else None), ln_addvec.stride(0) if (ln is not None and ln_addvec is not None) else 0,
            ln_rows_per_vec if ln is not None else 0, _p(yn), Cc if yn is not None else 0, _stream())
    if trace is not None:
        with trace.launch("ff_fused_c320", flops=flops, sig=f"ff_M{M}_C{Cc}_H{hidden}_r{int(bool(r32)) if R is not None else 'n'}_o{int(out.dtype == torch.float32)}{'_ln' if ln is not None else ''}", nbytes=nbytes):
            check(_lib.svd_ff_geglu_fused(*args), "svd_ff_geglu_fused")
        return out if ln is None else (out, yn)
    check(_lib.svd_ff_geglu_fused(*args), "svd_ff_geglu_fused")
    return out if ln is None else (out, yn)


# ROW-OWNING 320 -> 320 PROJECTION (round 6, csrc/rowgemm.hip): proj_in / attn1.to_out / proj_out of the 320-channel transformer blocks in ONE launch with the
# LayerNorm that follows (norm1 / norm3): a wave owns 32 token rows and all 320 outputs, the fp32 tensor the projection writes is not read back by a norm
# kernel.  SVD_ROWGEMM=0 keeps svd_gemm + svd_layernorm (A/B).
ROWGEMM = _os.environ.get("SVD_ROWGEMM", "1") != "0"
ROWGEMM_PLAIN_MIN_ROWS = int(_os.environ.get("SVD_ROWGEMM_PLAIN_MIN_ROWS", "200000"))     # projections WITHOUT a fused LayerNorm: rowgemm320 from this many rows on


def rowgemm_ok(x, w_img, rows_per_vec=0):
    """Can `rowgemm320` take this projection?  (320 -> 320, packed weights present, the per-frame vector constant inside a 32-row tile)"""
    return ROWGEMM and w_img is not None and x.shape[1] == 320 and (rows_per_vec == 0 or rows_per_vec % 32 == 0)


# ZERO-COPY CONCATENATION (round 6): the decoder half's torch.cat([h, hs.pop()], 1) (video_model.py:606-611) is a 16-bit GEMM operand.  Where the tensor a layer
# produces is consumed by that concatenation ONLY (a decoder block's output, an Upsample output, a CAM merger's output), the layer's last kernel writes its
# 16-bit rounding straight into its column range of the concatenation buffer instead of an fp32 tensor that svd_cast_rows_f32 rounds afterwards: the same bits
# (one rounding of the same fp32 value), 8 bytes per element less HBM traffic.  SVD_ZERO_COPY_CONCAT=0 restores the copies (A/B).
ZERO_COPY_CONCAT = _os.environ.get("SVD_ZERO_COPY_CONCAT", "1") != "0"
# ... also through svd_gemm's GENERIC epilogue (fp32 residual in, 16-bit rows out: proj_out of the 640 / 1280-channel transformers and mergers, the last temporal
# convolution of a ResBlock that ends a decoder block)
ZERO_COPY_GENERIC = _os.environ.get("SVD_ZERO_COPY_GENERIC", "1") != "0"          # measured: +0.3 % on the stage-1 line, identical results (profiles/r06_bench6_zero_copy_generic_ab.txt)


def rowgemm320(x, w_img, *, bias=None, rowvec=None, rows_per_vec=0, residual=None, out_f32=True, ln=None, eps=1e-5, want_y=True, out=None):
    """x [M, 320] 16-bit rows; w_img = video_model.pack_rowgemm320(W) on the device (uint8).  Returns (y, yn): y = residual + bias + rowvec[row // rows_per_vec]
    + x W^T (fp32 rows when out_f32, else 16 bit; None when want_y is False), yn = LayerNorm(y) * ln[0] + ln[1] in the 16-bit type (None without ln)."""
    assert x.dtype in (BF16, F16) and x.is_cuda and x.stride(1) == 1 and x.shape[1] == 320 and w_img.dtype == torch.uint8
    M = x.shape[0]
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(1) == 1 and residual.shape == x.shape
    if rowvec is not None:
        assert rowvec.dtype == torch.float32 and rowvec.stride(-1) == 1 and rows_per_vec % 32 == 0 and rows_per_vec > 0
    assert want_y or ln is not None
    if out is not None:          # a caller-owned (possibly strided) destination, e.g. a column range of a concatenation buffer
        assert want_y and out.shape == (M, 320) and out.stride(1) == 1 and out.dtype in (torch.float32, x.dtype) and out.is_cuda
        out_f32 = out.dtype == torch.float32
    y = out if out is not None else (torch.empty((M, 320), dtype=torch.float32 if out_f32 else x.dtype, device=x.device) if want_y else None)
    yn = torch.empty((M, 320), dtype=x.dtype, device=x.device) if ln is not None else None
    g, b = ln if ln is not None else (None, None)
    flops = 2.0 * M * 320 * 320
    nbytes = float(M * 320 * (2 + (4 if residual is not None else 0) + (y.element_size() if y is not None else 0) + (2 if yn is not None else 0)) + w_img.numel())
    if worklog is not None:
        _wl("rowgemm320_kernel", flops, nbytes)
    args = (_p(x), x.stride(0), _p(w_img), _p(bias), _p(rowvec), rowvec.stride(0) if rowvec is not None else 0, rows_per_vec, _p(residual),
            residual.stride(0) if residual is not None else 0, _p(y), y.stride(0) if y is not None else 0, int(out_f32), _p(g), _p(b), float(eps),
            _p(yn), yn.stride(0) if yn is not None else 0, M, _dt(x), _stream())
    if trace is not None:
        with trace.launch("rowgemm320", flops=flops, sig=f"rowgemm_M{M}_r{int(residual is not None)}_ln{int(ln is not None)}_o{int(out_f32)}", nbytes=nbytes):
            check(_lib.svd_rowgemm320(*args), "svd_rowgemm320")
        return y, yn
    check(_lib.svd_rowgemm320(*args), "svd_rowgemm320")
    return y, yn


# ROW-RESIDENT 320 -> N PROJECTION (round 6, csrc/rowproj.hip): the q | k and q | k | v projections of the 320-channel transformer blocks read their rows ONCE for
# the whole width (a wave keeps its 32 rows in registers; the 256 x 320 GEMM tile re-reads them per N tile).  SVD_ROWPROJ=0 keeps svd_gemm (A/B).
ROWPROJ = _os.environ.get("SVD_ROWPROJ", "1") != "0"
ROWPROJ_MIN_ROWS = int(_os.environ.get("SVD_ROWPROJ_MIN_ROWS", "65536"))


def rowproj_ok(x, w_img):
    return ROWPROJ and w_img is not None and x.shape[1] == 320 and x.shape[0] >= ROWPROJ_MIN_ROWS


def rowproj320(x, w_img, n_out, *, bias=None, out=None):
    """x [M, 320] 16-bit rows; w_img = video_model.pack_rowproj320(W [n_out, 320]) on the device (uint8).  Returns x W^T + bias, [M, n_out] in the 16-bit type."""
    assert x.dtype in (BF16, F16) and x.is_cuda and x.stride(1) == 1 and x.shape[1] == 320 and w_img.dtype == torch.uint8 and n_out % 64 == 0
    assert w_img.numel() == (n_out // 64) * 40960
    M = x.shape[0]
    if out is None:
        out = torch.empty((M, n_out), dtype=x.dtype, device=x.device)
    assert out.shape == (M, n_out) and out.stride(1) == 1 and out.dtype == x.dtype and out.stride(0) % 2 == 0
    flops = 2.0 * M * n_out * 320
    nbytes = float(M * (320 + n_out) * 2 + w_img.numel())
    if worklog is not None:
        _wl("rowproj320_kernel", flops, nbytes)
    args = (_p(x), x.stride(0), _p(w_img), _p(bias), _p(out), out.stride(0), M, n_out, _dt(x), _stream())
    if trace is not None:
        with trace.launch("rowproj320", flops=flops, sig=f"rowproj_M{M}_N{n_out}", nbytes=nbytes):
            check(_lib.svd_rowproj320(*args), "svd_rowproj320")
        return out
    check(_lib.svd_rowproj320(*args), "svd_rowproj320")
    return out


def attn_spatial(q, k, vt, out, frames, n_tok, heads):
    """q,k: views into a [frames*n_tok, ld] tensor at the head-0 column; vt: [frames, heads*64, tok_ld]."""
    if worklog is not None:
        _wl("attn_spatial_d64_kernel", 4.0 * frames * heads * n_tok * n_tok * 64, 4.0 * frames * n_tok * heads * 64 * 2)
    if trace is not None:
        with trace.launch("attn_spatial_d64", flops=4.0 * frames * heads * n_tok * n_tok * 64, sig=f"attn_spatial_f{frames}_n{n_tok}_h{heads}",
                          nbytes=4.0 * frames * n_tok * heads * 64 * 2):
            check(_lib.svd_attn_spatial_d64(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), vt.stride(1), _p(out),
                                            out.stride(0), frames, n_tok, heads, _dt(q), _stream()), "svd_attn_spatial_d64")
        return out
    check(_lib.svd_attn_spatial_d64(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), vt.stride(1), _p(out),
                                    out.stride(0), frames, n_tok, heads, _dt(q), _stream()), "svd_attn_spatial_d64")
    return out


def attn_temporal(q, k, v, out, batch, tq, tk, n_pix, heads):
    if worklog is not None:       # q, k, v read once + o written once; QK^T and PV flops
        _wl("attn_temporal", 4.0 * batch * n_pix * heads * tq * tk * 64, 2.0 * batch * n_pix * heads * 64 * (2 * tq + 2 * tk))
    check(_lib.svd_attn_temporal_d64(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(out),
                                     out.stride(0), batch, tq, tk, n_pix, heads, _dt(q), _stream()), "svd_attn_temporal_d64")
    return out


def softmax_rows(s, out, scale):
    rows, n = s.shape
    check(_lib.svd_softmax_rows(_p(s), s.stride(0), _p(out), out.stride(0), rows, n, float(scale), _dt(out), _stream()),
          "svd_softmax_rows")
    return out


def _gn_workspace(device, frames, channels, nstat=1, groups=32):
    """(partial sums, stats) workspaces of svd_groupnorm_*: partial >= svd_groupnorm_partial_elems, stats = [nstat][groups][2] floats."""
    need = int(_lib.svd_groupnorm_partial_elems(frames, channels))
    need_stats = 2 * groups * nstat
    ws = _gn_ws.get(device)
    if ws is None or ws[0].numel() < need or ws[1].numel() < need_stats:
        ws = (torch.empty(max(need, 1 << 20), dtype=torch.float32, device=device),
              torch.empty(max(need_stats, 65536), dtype=torch.float32, device=device))
        _gn_ws[device] = ws
    return ws


def groupnorm(x, frames, pix, gamma, beta, eps, *, frames_per_stat=1, silu=False, groups=32, out=None):
    """GroupNorm(32) (+SiLU) on a [frames*pix, C] token tensor; statistics over frames_per_stat frames."""
    rows, ld = _rows_ld(x)
    Cc = x.shape[1]
    assert rows == frames * pix
    assert frames % frames_per_stat == 0
    partial, stats = _gn_workspace(x.device, frames, Cc, frames // frames_per_stat, groups)
    if worklog is not None:
        _wl("gn_stats_partial_kernel", 0.0, float(rows) * Cc * x.element_size())
        _wl("gn_apply_kernel", 0.0, float(rows) * Cc * (x.element_size() + 2))
    if out is None:
        out = torch.empty((rows, Cc), dtype=_odt(x), device=x.device)
    # one call: statistics pass + apply pass (the apply pass finalizes per-frame statistics itself: no gn_finalize launch)
    check(_lib.svd_groupnorm(_p(x), ld, _p(out), out.stride(0), frames, pix, Cc, groups, frames_per_stat, float(eps), _p(partial),
                             _p(stats), _p(gamma), _p(beta), int(silu), _dt_in(x), _stream()), "svd_groupnorm")
    return out


def groupnorm_sums(x, frames, pix, frames_per_stat, groups=32):
    """This rank's share of the GroupNorm statistics: [frames/frames_per_stat, groups, 2] float64 (sum, sum of squares) over its rows;
    the ranks of a sequence-parallel group add them (parallel.SeqParallel.allreduce_sums) before groupnorm_apply_sums."""
    rows, ld = _rows_ld(x)
    Cc = x.shape[1]
    assert rows == frames * pix and frames % frames_per_stat == 0
    partial, _ = _gn_workspace(x.device, frames, Cc, frames // frames_per_stat, groups)
    sums = torch.empty((frames // frames_per_stat, groups, 2), dtype=torch.float64, device=x.device)
    if worklog is not None:
        _wl("gn_stats_partial_kernel", 0.0, float(rows) * Cc * x.element_size())
    check(_lib.svd_groupnorm_sums(_p(x), ld, frames, pix, Cc, groups, frames_per_stat, _p(partial), _p(sums), _dt_in(x), _stream()),
          "svd_groupnorm_sums")
    return sums


def groupnorm_apply_sums(x, frames, pix, gamma, beta, eps, sums, count, *, frames_per_stat=1, silu=False, groups=32, out=None):
    """GroupNorm (+SiLU) of the local rows with statistics formed from (all-reduced) sums over `count` elements per (batch, group)."""
    rows, ld = _rows_ld(x)
    Cc = x.shape[1]
    nstat = frames // frames_per_stat
    assert sums.dtype == torch.float64 and sums.is_contiguous() and tuple(sums.shape) == (nstat, groups, 2)
    _, stats = _gn_workspace(x.device, frames, Cc, nstat, groups)
    if worklog is not None:
        _wl("gn_apply_kernel", 0.0, float(rows) * Cc * (x.element_size() + 2))
    check(_lib.svd_groupnorm_stats_from_sums(_p(sums), nstat, groups, float(count), float(eps), _p(stats), _stream()),
          "svd_groupnorm_stats_from_sums")
    if out is None:
        out = torch.empty((rows, Cc), dtype=_odt(x), device=x.device)
    check(_lib.svd_groupnorm_apply(_p(x), ld, _p(out), out.stride(0), frames, pix, Cc, groups, frames_per_stat,
                                   _p(stats), _p(gamma), _p(beta), int(silu), _dt_in(x), _stream()), "svd_groupnorm_apply")
    return out


def layernorm(x, gamma, beta, *, eps=1e-5, addvec=None, rows_per_vec=0, want_sum=False, silu=False, out=None):
    rows, ld = _rows_ld(x)
    Cc = x.shape[1]
    if out is None:
        out = torch.empty((rows, Cc), dtype=_odt(x), device=x.device)
    xsum = torch.empty((rows, Cc), dtype=x.dtype, device=x.device) if want_sum else None      # fp32 stream in -> fp32 sum out (it continues the stream)
    if worklog is not None:
        _wl("layernorm_kernel", 0.0, float(rows) * Cc * (x.element_size() * (2 if want_sum else 1) + 2))
    check(_lib.svd_layernorm(_p(x), ld, _p(out), out.stride(0), rows, Cc, _p(gamma), _p(beta), float(eps),
                             _p(addvec), addvec.stride(0) if addvec is not None else 0, rows_per_vec,
                             _p(xsum), xsum.stride(0) if xsum is not None else 0, int(silu), _dt_in(x), _stream()),
          "svd_layernorm")
    return (out, xsum) if want_sum else out


def nchw_to_tokens(x0, x1, scale, cpad):
    """[F,c0,H,W] (+ [F,c1,H,W]) fp32 -> [F*H*W, cpad] bf16 (x0 scaled per frame by scale[F])."""
    F_, c0 = x0.shape[0], x0.shape[1]
    pix = x0.shape[2] * x0.shape[3]
    c1 = x1.shape[1] if x1 is not None else 0
    assert x0.dtype == torch.float32 and x0.is_contiguous()
    if x1 is not None:
        assert x1.dtype == torch.float32 and x1.is_contiguous() and x1.shape[0] == F_
    out = torch.empty((F_ * pix, cpad), dtype=ELEM, device=x0.device)
    check(_lib.svd_nchw_to_tokens(_p(x0), c0, _p(x1), c1, _p(scale), _p(out), cpad, F_, pix, _dt(out), _stream()),
          "svd_nchw_to_tokens")
    return out


def nchw_to_tokens_x3(x0, x1, scale, cpad):
    """nchw_to_tokens with SPLIT-3 rows [F*H*W, 3*cpad] = [hi | lo | hi] (extended-precision operand of the stem convolutions)."""
    F_, c0 = x0.shape[0], x0.shape[1]
    pix = x0.shape[2] * x0.shape[3]
    c1 = x1.shape[1] if x1 is not None else 0
    assert x0.dtype == torch.float32 and x0.is_contiguous()
    if x1 is not None:
        assert x1.dtype == torch.float32 and x1.is_contiguous() and x1.shape[0] == F_
    out = torch.empty((F_ * pix, 3 * cpad), dtype=ELEM, device=x0.device)
    check(_lib.svd_nchw_to_tokens_x3(_p(x0), c0, _p(x1), c1, _p(scale), _p(out), cpad, F_, pix, _dt(out), _stream()),
          "svd_nchw_to_tokens_x3")
    return out


def rows_split3(x, ln=None, eps=1e-5, silu=False):
    """fp32 rows [rows, C] -> split-3 16-bit rows [rows, 3 C]; ln = (gamma, beta): per-row LayerNorm first (C <= 512), then SiLU (both fp32)."""
    rows, ld = _rows_ld(x)
    assert x.dtype == torch.float32
    Cc = x.shape[1]
    out = torch.empty((rows, 3 * Cc), dtype=ELEM, device=x.device)
    flags = (1 if ln is not None else 0) | (2 if silu else 0)
    g, b = ln if ln is not None else (None, None)
    check(_lib.svd_rows_split3(_p(x), ld, _p(out), out.stride(0), rows, Cc, _p(g), _p(b), float(eps), flags, _dt(out), _stream()),
          "svd_rows_split3")
    return out


def add_rows_f32b(x, b):
    """x + b row-wise with b fp32 (x 16 bit or the fp32 stream; the sum has x's type, rounded once)."""
    assert b.dtype == torch.float32 and b.shape == x.shape and b.stride(1) == 1
    out = torch.empty_like(x)
    dt = (_DT[ELEM] | _l.DTYPE_IN_F32) if x.dtype == torch.float32 else _dt(x)
    check(_lib.svd_add_rows_bf32(_p(x), x.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), x.shape[0], x.shape[1], dt, _stream()),
          "svd_add_rows_bf32")
    return out


def head_gn_silu_conv3x3(x, frames, H, W, gamma, beta, eps, wt, bias, cout, groups=32):
    """conv3x3(SiLU(GroupNorm(x))) + bias -> fp32 [frames*H*W, 4] (first `cout` columns valid), all in fp32 arithmetic: the UNet's head."""
    rows, ld = _rows_ld(x)
    Cc = x.shape[1]
    pix = H * W
    assert rows == frames * pix and wt.dtype == torch.float32 and wt.is_contiguous() and tuple(wt.shape) == (9, Cc, 4)
    partial, stats = _gn_workspace(x.device, frames, Cc, frames, groups)
    if worklog is not None:
        _wl("gn_stats_partial_kernel", 0.0, float(rows) * Cc * x.element_size())
        _wl("head_gn_silu_conv3x3_kernel", 2.0 * rows * 9 * Cc * 4, float(rows) * Cc * x.element_size() + rows * 16.0)
    check(_lib.svd_groupnorm_stats(_p(x), ld, frames, pix, Cc, groups, 1, float(eps), _p(partial), _p(stats), _dt_in(x), _stream()),
          "svd_groupnorm_stats")
    out = torch.empty((rows, 4), dtype=torch.float32, device=x.device)
    check(_lib.svd_head_gn_silu_conv3x3(_p(x), ld, frames, H, W, Cc, groups, 1, _p(stats), _p(gamma), _p(beta), _p(wt), _p(bias), _p(out), 4, cout,
                                        _dt_in(x), _stream()), "svd_head_gn_silu_conv3x3")
    return out


def tokens_to_nchw(x, c, frames, h, w):
    out = torch.empty((frames, c, h, w), dtype=torch.float32, device=x.device)
    check(_lib.svd_tokens_to_nchw(_p(x), _dt(x), x.stride(0), _p(out), c, frames, h * w,
                                  _stream()), "svd_tokens_to_nchw")
    return out


def concat_channels(a, b):
    """torch.cat((a, b), channels) -> 16-bit [rows, ca + cb].  fp32 inputs (the residual stream and the encoder's skip tensors) are rounded
    to the element type on the way: the concatenation is a GEMM operand (skip_connection) and a GroupNorm input."""
    rows = a.shape[0]
    if worklog is not None and a.dtype != torch.float32 and b.dtype != torch.float32:
        _wl("copy_rows_kernel", 0.0, 4.0 * rows * (a.shape[1] + b.shape[1]))
        _wl("copy_rows_kernel", 0.0, 0.0)            # two launches
    if a.dtype == torch.float32 or b.dtype == torch.float32:
        out = torch.empty((rows, a.shape[1] + b.shape[1]), dtype=ELEM, device=a.device)
        to_elem_rows(a, out=out[:, :a.shape[1]])
        to_elem_rows(b, out=out[:, a.shape[1]:])
        return out
    out = torch.empty((rows, a.shape[1] + b.shape[1]), dtype=a.dtype, device=a.device)
    check(_lib.svd_concat_channels(_p(a), a.stride(0), a.shape[1], _p(b), b.stride(0), b.shape[1], _p(out),
                                   out.stride(0), rows, _stream()), "svd_concat_channels")
    return out


def to_elem_rows(x, out=None):
    """[rows, C] fp32 (row stride free) -> 16-bit element type, vectorised; 16-bit input: copied into `out` or returned as is."""
    rows, ld = _rows_ld(x)
    if x.dtype != torch.float32:
        if out is None:
            return x
        out.copy_(x)
        return out
    if out is None:
        out = torch.empty((rows, x.shape[1]), dtype=ELEM, device=x.device)
    assert out.shape == x.shape and out.stride(1) == 1
    if worklog is not None:
        _wl("cast_rows_f32_kernel", 0.0, 6.0 * rows * x.shape[1])
    check(_lib.svd_cast_rows_f32(_p(x), ld, _p(out), out.stride(0), rows, x.shape[1], _dt(out), _stream()), "svd_cast_rows_f32")
    return out


def add_rows(x, b):
    """x + b row-wise; x may be the fp32 residual stream (then the sum is fp32), b is 16 bit."""
    out = torch.empty_like(x)
    if worklog is not None:
        _wl("add_rows_f32_kernel" if x.dtype == torch.float32 else "add_rows_kernel", 0.0, float(x.numel()) * (2 * x.element_size() + 2))
    dt = (_dt(b) | _l.DTYPE_IN_F32) if x.dtype == torch.float32 else _dt(x)
    check(_lib.svd_add_rows(_p(x), x.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), x.shape[0], x.shape[1],
                            dt, _stream()), "svd_add_rows")
    return out


def permute_rows(x, dims, perm, out=None):
    """x viewed as rows [d0, d1, d2, d3, C] -> the same rows ordered [d_perm0, d_perm1, d_perm2, d_perm3, C] (one pass, 16-byte vectors)."""
    assert x.is_contiguous() and len(dims) == 4 and len(perm) == 4
    n = dims[0] * dims[1] * dims[2] * dims[3]
    assert x.shape[0] == n and (x.shape[1] * x.element_size()) % 16 == 0
    if out is None:
        out = torch.empty_like(x)
    check(_lib.svd_permute_rows(_p(x), _p(out), *[int(d) for d in dims], *[int(q) for q in perm], x.shape[1] * x.element_size(), _stream()),
          "svd_permute_rows")
    return out


def to_elem(x, silu=False):
    """fp32 -> ELEM (optionally through SiLU) -- the only 'cast' kernel; x any shape, contiguous."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=ELEM, device=x.device)
    check(_lib.svd_cast_f32(_p(x), _p(out), x.numel(), int(silu), _dt(out), _stream()), "svd_cast_f32")
    return out


to_bf16 = to_elem   # historical name


def timestep_embedding(t, dim, max_period=10000.0, f32=False):
    assert t.dtype == torch.float32 and t.is_contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.float32 if f32 else ELEM, device=t.device)
    check(_lib.svd_timestep_embedding(_p(t), t.numel(), dim, float(max_period), _p(out), _dt(out), _stream()),
          "svd_timestep_embedding")
    return out


def edm_euler_step(x, net, guidance_scale, sigma, sigma_next):
    """In-place Euler-EDM step on x [T,C,H,W] fp32 given the raw network output net [2T*pix, ldn] fp32."""
    T, Cc = x.shape[0], x.shape[1]
    pix = x.shape[2] * x.shape[3]
    check(_lib.svd_edm_euler_step(_p(x), _p(net), net.stride(0), _p(guidance_scale), T, Cc, pix, float(sigma),
                                  float(sigma_next), _stream()), "svd_edm_euler_step")
    return x


def ae_time_mix3(x, w, b, frames, h, wd, clamp):
    out = torch.empty((frames, 3, h, wd), dtype=torch.float32, device=x.device)
    check(_lib.svd_ae_time_mix3(_p(x), x.stride(0), _p(w), _p(b), _p(out), frames, h * wd, int(clamp), _stream()),
          "svd_ae_time_mix3")
    return out


# ---- I2VGen-XL enhancement stage (row A12) ---------------------------------------------------------------------------
def attn_cross(q, k, vt, out, frames, n_q, n_k, frames_per_kv, heads):
    """q: view at head 0 of [frames*n_q, ld]; k: [(frames/frames_per_kv)*n_k, ld]; vt: [frames/frames_per_kv, heads*64, tok_ld]."""
    if worklog is not None:
        _wl("attn_spatial_d64_kernel", 4.0 * frames * heads * n_q * n_k * 64, 2.0 * frames * n_q * heads * 64 * 2)
    args = (_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), vt.stride(1), _p(out), out.stride(0), frames, n_q, n_k,
            frames_per_kv, heads, _dt(q), _stream())
    if trace is not None:
        with trace.launch("attn_cross_d64", flops=4.0 * frames * heads * n_q * n_k * 64):
            check(_lib.svd_attn_cross_d64(*args), "svd_attn_cross_d64")
        return out
    check(_lib.svd_attn_cross_d64(*args), "svd_attn_cross_d64")
    return out


def adaptive_avgpool(x, frames, hin, win, hout, wout):
    Cc = x.shape[1]
    out = torch.empty((frames * hout * wout, Cc), dtype=x.dtype, device=x.device)
    check(_lib.svd_adaptive_avgpool_tokens(_p(x), x.stride(0), _p(out), out.stride(0), frames, hin, win, hout, wout, Cc, _dt(x),
                                           _stream()), "svd_adaptive_avgpool_tokens")
    return out


def i2v_image_temporal_encoder(x, params, batch, frames, h, w):
    """x [(b f) h w, >=4] tokens (16 bit, or fp32 rows) -> fp32 [(b f), 4, h, w]."""
    assert params.dtype == torch.float32 and params.numel() == 288 and params.is_contiguous()
    out = torch.empty((batch * frames, 4, h, w), dtype=torch.float32, device=x.device)
    check(_lib.svd_i2v_image_temporal_encoder(_p(x), x.stride(0), _p(params), _p(out), batch, frames, h * w, _dt_in(x), _stream()),
          "svd_i2v_image_temporal_encoder")
    return out


def ddim_cfg_step(x, pred_uncond, pred_cond, guidance_scale, alpha_t, alpha_prev, v_prediction=True, out=None):
    """fp32 tensors of equal shape (contiguous); returns x_prev (diffusers DDIMScheduler.step, eta 0, after CFG)."""
    for t in (x, pred_uncond) + ((pred_cond,) if pred_cond is not None else ()):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == x.numel()
    if out is None:
        out = torch.empty_like(x)
    check(_lib.svd_ddim_cfg_step(_p(x), _p(pred_uncond), _p(pred_cond), _p(out), x.numel(), float(guidance_scale), float(alpha_t),
                                 float(alpha_prev), int(v_prediction), _stream()), "svd_ddim_cfg_step")
    return out


def frames_to_uint8(frames):
    """fp32 [F, 3, H, W] in [-1, 1] -> uint8 [F, H, W, 3]; bit-exact with the reference's convert_range + IImage (row A13)."""
    assert frames.dtype == torch.float32 and frames.is_contiguous() and frames.shape[1] == 3
    F_, _, H, W = frames.shape
    out = torch.empty((F_, H, W, 3), dtype=torch.uint8, device=frames.device)
    check(_lib.svd_frames_to_uint8(_p(frames), _p(out), F_, H * W, _stream()), "svd_frames_to_uint8")
    return out


def gelu_(x):
    """exact-erf GELU in place on a [rows, C] 16-bit token tensor."""
    rows, ld = _rows_ld(x)
    check(_lib.svd_gelu_rows(_p(x), ld, rows, x.shape[1], _dt(x), _stream()), "svd_gelu_rows")
    return x


# ---- EMA-VFI operators (csrc/vfi.hip) ---------------------------------------------------------------------------------------------
def prelu_(x, slope):
    """nn.PReLU in place on a [rows, C] 16-bit token tensor; slope fp32 [C]."""
    rows, ld = _rows_ld(x)
    assert slope.dtype == torch.float32 and slope.numel() >= x.shape[1]
    check(_lib.svd_prelu_rows(_p(x), ld, rows, x.shape[1], _p(slope), _dt(x), _stream()), "svd_prelu_rows")
    return x


def dwconv3x3_gelu(x, w9, bias, frames, h, w):
    """depthwise 3x3 (padding 1) + bias + exact GELU on [frames*h*w, C] tokens; w9 fp32 [9, C]."""
    rows, ld = _rows_ld(x)
    assert rows == frames * h * w and w9.dtype == torch.float32 and bias.dtype == torch.float32
    out = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
    check(_lib.svd_dwconv3x3_gelu(_p(x), ld, _p(out), out.stride(0), _p(w9), _p(bias), frames, h, w, x.shape[1], _dt(x), _stream()),
          "svd_dwconv3x3_gelu")
    return out


def window_attn_7x7(q, kv, ce, mask, n_win, heads, motion_per_head, scale):
    """q [n_win*49, >=heads*32], kv [n_win*49, >=2*heads*32] 16-bit; ce fp32 [n_win*49, >=heads*mph]; mask fp32 [nW, 49, 49] or None.
    Returns (x [n_win*49, heads*32], c_reverse - cor_embed [n_win*49, heads*mph]) in the element type."""
    assert q.shape[0] == n_win * 49 and kv.shape[0] == n_win * 49 and ce.shape[0] == n_win * 49 and ce.dtype == torch.float32
    ox = torch.empty((n_win * 49, heads * 32), dtype=q.dtype, device=q.device)
    oc = torch.empty((n_win * 49, heads * motion_per_head), dtype=q.dtype, device=q.device)
    if mask is not None:
        assert mask.dtype == torch.float32 and mask.is_contiguous() and tuple(mask.shape[1:]) == (49, 49)
    check(_lib.svd_window_attn_7x7(_p(q), q.stride(0), _p(kv), kv.stride(0), _p(ce), ce.stride(0), _p(mask), mask.shape[0] if mask is not None else 0,
                                   _p(ox), ox.stride(0), _p(oc), oc.stride(0), n_win, heads, motion_per_head, float(scale), _dt(q), _stream()),
          "svd_window_attn_7x7")
    return ox, oc


def warp_bilinear(x, flow, frames, h, w):
    """backward warp of channels-last x [frames*h*w, C] (fp32 or 16-bit) by flow: fp32 view [frames*h*w, 2] (row stride arbitrary)."""
    rows, ld = _rows_ld(x)
    assert rows == frames * h * w and flow.dtype == torch.float32 and flow.shape == (rows, 2) and flow.stride(1) == 1
    out = torch.empty((rows, x.shape[1]), dtype=x.dtype, device=x.device)
    check(_lib.svd_warp_bilinear(_p(x), ld, _p(out), out.stride(0), _p(flow), flow.stride(0), frames, h, w, x.shape[1], _dt(x), _stream()),
          "svd_warp_bilinear")
    return out


def resize_bilinear(x, frames, hin, win, scale_factor, *, mult=None, out=None, accumulate=False):
    """F.interpolate(bilinear, align_corners=False, scale_factor) on channels-last fp32 x [frames*hin*win, C] -> ([frames*hout*wout, C], hout, wout);
    out (+)= mult[c] * value."""
    rows, ld = _rows_ld(x)
    assert x.dtype == torch.float32 and rows == frames * hin * win
    hout, wout = int(hin * scale_factor), int(win * scale_factor)
    Cc = x.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty((frames * hout * wout, Cc), dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.shape[0] == frames * hout * wout and out.stride(1) == 1 and out.shape[1] >= Cc
    if mult is not None:
        assert mult.dtype == torch.float32 and mult.numel() >= Cc
    check(_lib.svd_resize_bilinear_f32(_p(x), ld, _p(out), out.stride(0), frames, hin, win, hout, wout, Cc, float(scale_factor), _p(mult),
                                       int(accumulate), _stream()), "svd_resize_bilinear_f32")
    return out, hout, wout


def vfi_merge(warped0, warped1, mask, unet_out, want_merged=False):
    """fp32 [n, 3] x2, mask view [n, 1], unet_out [n, >=3] (pre-sigmoid) -> pred [n, 3] (and merged)."""
    n = warped0.shape[0]
    assert all(t.dtype == torch.float32 for t in (warped0, warped1, mask, unet_out)) and warped0.is_contiguous() and warped1.is_contiguous()
    assert mask.shape[0] == n and unet_out.shape[0] == n and unet_out.stride(1) == 1
    pred = torch.empty((n, 3), dtype=torch.float32, device=warped0.device)
    merged = torch.empty_like(pred) if want_merged else None
    check(_lib.svd_vfi_merge(_p(warped0), _p(warped1), _p(mask), mask.stride(0), _p(unet_out), unet_out.stride(0), _p(merged), _p(pred), n,
                             _stream()), "svd_vfi_merge")
    return (pred, merged) if want_merged else pred


def vfi_tta_average(pred2, h, w, want_uint8=False):
    """pred2 fp32 [2*h*w, 3] (the pair and its 180-degree rotation) -> (average fp32 [h*w, 3], uint8 [h, w, 3] or None)."""
    assert pred2.dtype == torch.float32 and pred2.is_contiguous() and pred2.shape == (2 * h * w, 3)
    out = torch.empty((h * w, 3), dtype=torch.float32, device=pred2.device)
    u8 = torch.empty((h, w, 3), dtype=torch.uint8, device=pred2.device) if want_uint8 else None
    check(_lib.svd_vfi_tta_average(_p(pred2), _p(out), _p(u8), h, w, _stream()), "svd_vfi_tta_average")
    return out, u8
