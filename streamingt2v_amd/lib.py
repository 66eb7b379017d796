"""ctypes binding of libsvdhip.so (the C ABI declared in include/svdhip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make -C streamingt2v_amd/csrc``.
There is NO fallback: if the library is missing or a symbol is absent the import fails loudly -- the product
path never silently degrades to PyTorch ops or to the CPU oracle.
"""
import ctypes as C
import os

# torch FIRST: it ships its own libamdhip64.so and the host side hands torch's streams and device pointers to this library.  Loading
# libsvdhip.so before torch binds it to the system /opt/rocm HIP runtime instead -- a second runtime in the process, whose first launch
# fails with "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))


def _lib_file():
    """libsvdhip.so, or a developer build next to it (reduced tile table / timing probes): SVD_LIB=probe|probe2, or SVD_LIB_FILE=<basename>.
    Only a plain file name of the form libsvdhip*.so INSIDE the package directory is accepted -- the variable can never make the package
    dlopen an arbitrary path."""
    import re
    name = {"probe": "libsvdhip_probe.so", "probe2": "libsvdhip_probe2.so"}.get(os.environ.get("SVD_LIB", ""))
    if name is None:
        name = os.environ.get("SVD_LIB_FILE", "libsvdhip.so")
        if not re.fullmatch(r"libsvdhip[A-Za-z0-9_.-]*\.so", name) or os.path.basename(name) != name or ".." in name:
            raise ImportError(f"SVD_LIB_FILE={name!r}: only a file name matching libsvdhip*.so inside {_HERE} is accepted")
    return name


LIB_PATH = os.path.join(_HERE, _lib_file())

ABI_VERSION = 11

# entry points include/svdhip.h declares (checked at load; tests/test_abi.py re-checks against the header text)
SYMBOLS = [
    "svd_abi_version", "svd_last_error", "svd_gemm", "svd_gemm_num_configs", "svd_gemm_config_info", "svd_gemm_pick_config", "svd_gemm_config_valid",
    "svd_attn_spatial_d64", "svd_attn_temporal_d64", "svd_softmax_rows",
    "svd_groupnorm_partial_elems", "svd_groupnorm_stats", "svd_groupnorm_apply", "svd_groupnorm", "svd_groupnorm_sums", "svd_groupnorm_stats_from_sums", "svd_layernorm",
    "svd_nchw_to_tokens", "svd_nchw_to_tokens_x3", "svd_rows_split3", "svd_add_rows_bf32", "svd_head_gn_silu_conv3x3", "svd_tokens_to_nchw", "svd_concat_channels", "svd_add_rows", "svd_cast_f32", "svd_cast_rows_f32", "svd_permute_rows",
    "svd_timestep_embedding", "svd_edm_euler_step", "svd_ae_time_mix3",
    "svd_attn_cross_d64", "svd_adaptive_avgpool_tokens", "svd_i2v_image_temporal_encoder", "svd_ddim_cfg_step", "svd_frames_to_uint8", "svd_gelu_rows",
    "svd_ff_geglu_fused", "svd_ff_fused_pack_bytes", "svd_rowgemm320", "svd_rowgemm320_pack_bytes", "svd_rowproj320", "svd_rowproj320_pack_bytes",
    "svd_prelu_rows", "svd_dwconv3x3_gelu", "svd_window_attn_7x7", "svd_warp_bilinear", "svd_resize_bilinear_f32", "svd_vfi_merge", "svd_vfi_tta_average",
]

A_PLAIN, A_CONV3X3, A_TEMPORAL3 = 0, 1, 2
OUT_BF16, OUT_F32, OUT_BF16_T = 0, 1, 2
EPI_GEGLU = 1
EPI_SILU = 2
DTYPE_BF16, DTYPE_F16, DTYPE_F32 = 0, 1, 2
DTYPE_IN_F32 = 0x100      # OR-ed into a 16-bit dtype: the input of a norm / add_rows is the fp32 residual stream


class GemmArgs(C.Structure):
    """Mirror of ``struct svd_gemm_args`` (field order and types must match include/svdhip.h)."""
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p), ("ldw", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("a_mode", C.c_int32),
        ("cin", C.c_int32),
        ("hin", C.c_int32), ("win", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32),
        ("stride", C.c_int32), ("ups", C.c_int32),
        ("t_frames", C.c_int32), ("rows_per_frame", C.c_int32),
        ("zeros", C.c_void_p),
        ("bias", C.c_void_p),
        ("rowvec", C.c_void_p), ("rowvec_ld", C.c_int32), ("rows_per_vec", C.c_int32),
        ("R", C.c_void_p), ("ldr", C.c_int64),
        ("S", C.c_void_p), ("lds", C.c_int64),
        ("alpha", C.c_float),
        ("epi_flags", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("out_mode", C.c_int32),
        ("tok_per_frame", C.c_int32), ("tokens_ld", C.c_int64),
        ("tile_cfg", C.c_int32),
        ("dtype", C.c_int32),
        ("dbg_cycles", C.c_void_p),
        ("pad_mode", C.c_int32),
        ("res_f32", C.c_int32),
    ]


class SvdHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C streamingt2v_amd/csrc). There is no non-HIP fallback.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"libsvdhip.so lacks symbols declared in include/svdhip.h: {missing}")
    lib.svd_abi_version.restype = C.c_int
    lib.svd_last_error.restype = C.c_char_p
    lib.svd_groupnorm_partial_elems.restype = C.c_int64
    lib.svd_groupnorm_partial_elems.argtypes = [C.c_int32, C.c_int32]
    i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float
    lib.svd_gemm.argtypes = [C.POINTER(GemmArgs), vp]
    lib.svd_gemm_pick_config.argtypes = [C.POINTER(GemmArgs)]
    lib.svd_gemm_config_valid.argtypes = [C.POINTER(GemmArgs), C.c_int]
    lib.svd_gemm_config_info.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 4
    lib.svd_attn_spatial_d64.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp]
    lib.svd_attn_temporal_d64.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp]
    lib.svd_softmax_rows.argtypes = [vp, i64, vp, i64, i64, i32, f32, i32, vp]
    lib.svd_groupnorm_stats.argtypes = [vp, i64, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp]
    lib.svd_groupnorm_apply.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp]
    lib.svd_groupnorm.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, i32, i32, vp]
    lib.svd_groupnorm_sums.argtypes = [vp, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp]
    lib.svd_groupnorm_stats_from_sums.argtypes = [vp, i32, i32, C.c_double, f32, vp, vp]
    lib.svd_layernorm.argtypes = [vp, i64, vp, i64, i64, i32, vp, vp, f32, vp, i32, i32, vp, i64, i32, i32, vp]
    lib.svd_nchw_to_tokens.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.svd_tokens_to_nchw.argtypes = [vp, i32, i64, vp, i32, i32, i32, vp]
    lib.svd_concat_channels.argtypes = [vp, i64, i32, vp, i64, i32, vp, i64, i64, vp]
    lib.svd_nchw_to_tokens_x3.argtypes = [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp]
    lib.svd_rows_split3.argtypes = [vp, i64, vp, i64, i64, i32, vp, vp, f32, i32, i32, vp]
    lib.svd_add_rows_bf32.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp]
    lib.svd_head_gn_silu_conv3x3.argtypes = [vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    lib.svd_add_rows.argtypes = [vp, i64, vp, i64, vp, i64, i64, i32, i32, vp]
    lib.svd_cast_f32.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.svd_cast_rows_f32.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp]
    lib.svd_permute_rows.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i64, vp]
    lib.svd_timestep_embedding.argtypes = [vp, i32, i32, f32, vp, i32, vp]
    lib.svd_edm_euler_step.argtypes = [vp, vp, i64, vp, i32, i32, i32, f32, f32, vp]
    lib.svd_ae_time_mix3.argtypes = [vp, i64, vp, vp, vp, i32, i32, i32, vp]
    lib.svd_attn_cross_d64.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp]
    lib.svd_adaptive_avgpool_tokens.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.svd_i2v_image_temporal_encoder.argtypes = [vp, i64, vp, vp, i32, i32, i32, i32, vp]
    lib.svd_ddim_cfg_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, i32, vp]
    lib.svd_frames_to_uint8.argtypes = [vp, vp, i32, i32, vp]
    lib.svd_gelu_rows.argtypes = [vp, i64, i64, i32, i32, vp]
    lib.svd_ff_fused_pack_bytes.restype = C.c_int64
    lib.svd_ff_fused_pack_bytes.argtypes = [i32]
    lib.svd_ff_geglu_fused.argtypes = [vp, i64, vp, i32, i32, vp, vp, i64, vp, i64, f32, i32, vp, i64, i32, i64, i32, vp, i32, i32,
                                       vp, vp, f32, vp, i32, i32, vp, i64, vp]
    lib.svd_rowgemm320_pack_bytes.restype = C.c_int64
    lib.svd_rowgemm320_pack_bytes.argtypes = []
    lib.svd_rowgemm320.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, i64, vp, i64, i32, vp, vp, f32, vp, i64, i64, i32, vp]
    lib.svd_rowproj320_pack_bytes.restype = C.c_int64
    lib.svd_rowproj320_pack_bytes.argtypes = [i32]
    lib.svd_rowproj320.argtypes = [vp, i64, vp, vp, vp, i64, i64, i32, i32, vp]
    lib.svd_prelu_rows.argtypes = [vp, i64, i64, i32, vp, i32, vp]
    lib.svd_dwconv3x3_gelu.argtypes = [vp, i64, vp, i64, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.svd_window_attn_7x7.argtypes = [vp, i64, vp, i64, vp, i64, vp, i32, vp, i64, vp, i64, i32, i32, i32, f32, i32, vp]
    lib.svd_warp_bilinear.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.svd_resize_bilinear_f32.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp]
    lib.svd_vfi_merge.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, i64, vp]
    lib.svd_vfi_tta_average.argtypes = [vp, vp, vp, i32, i32, vp]
    for s in SYMBOLS:
        fn = getattr(lib, s)
        if s not in ("svd_last_error", "svd_groupnorm_partial_elems", "svd_ff_fused_pack_bytes", "svd_rowgemm320_pack_bytes", "svd_rowproj320_pack_bytes"):
            fn.restype = C.c_int
    if lib.svd_abi_version() != ABI_VERSION:
        raise ImportError(f"libsvdhip.so ABI {lib.svd_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    return lib


lib = _load()


_DEBUG_SYNC = bool(os.environ.get("SVD_DEBUG_SYNC"))    # developer aid: device-synchronise after every launch (a faulting kernel then
                                                          # aborts inside ITS OWN launcher call: the Python traceback names it)


def check(rc, what):
    if _DEBUG_SYNC and rc == 0:
        torch.cuda.synchronize()
    if rc != 0:
        msg = lib.svd_last_error().decode() if rc == -2 else "invalid argument (SVD_EINVAL)"
        raise SvdHipError(f"{what} failed with code {rc}: {msg}")
