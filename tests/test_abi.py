"""CPU suite, part 2: the C-ABI library loads without a GPU and exports every symbol include/svdhip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "svdhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svd_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from streamingt2v_amd import lib
    names = _declared()
    assert len(names) >= 20
    assert sorted(lib.SYMBOLS) == names, "binding list out of sync with include/svdhip.h"
    dll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), f"{n} declared in svdhip.h but not exported"
    assert dll.svd_abi_version() == lib.ABI_VERSION


def test_gemm_args_struct_matches_header():
    """Field order of the ctypes mirror == field order of struct svd_gemm_args."""
    from streamingt2v_amd.lib import GemmArgs
    txt = open(os.path.join(ROOT, "include", "svdhip.h")).read()
    body = re.search(r"typedef struct svd_gemm_args \{(.*?)\} svd_gemm_args;", txt, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
    assert fields == [f[0] for f in GemmArgs._fields_]


def test_argument_validation_without_gpu():
    """Bad arguments are rejected with SVD_EINVAL before any device work (no GPU needed)."""
    from streamingt2v_amd.lib import GemmArgs, lib
    a = GemmArgs()
    assert lib.svd_gemm(ctypes.byref(a), None) == -1
    assert lib.svd_attn_spatial_d64(None, 0, None, 0, None, 0, None, 0, 1, 1, 1, 0, None) == -1
    assert lib.svd_attn_temporal_d64(None, 0, None, 0, None, 0, None, 0, 1, 129, 1, 1, 1, 0, None) == -1
    bm, bn, th, lds = (ctypes.c_int() for _ in range(4))
    assert lib.svd_gemm_config_info(1, bm, bn, th, lds) == 0 and (bm.value, bn.value) == (128, 128)
    assert lib.svd_gemm_config_info(99, bm, bn, th, lds) == -1
    # round-6 entry points: widths, alignments and option combinations are checked before the device is touched
    buf = (ctypes.c_char * 4096)()
    a16 = ctypes.cast(ctypes.addressof(buf) + (-ctypes.addressof(buf)) % 64, ctypes.c_void_p)          # a 64-byte aligned host address (never dereferenced)
    assert lib.svd_rowproj320_pack_bytes(960) == 15 * 40960 and lib.svd_rowproj320_pack_bytes(100) == -1
    assert lib.svd_rowproj320(a16, 320, a16, None, a16, 960, 128, 100, 1, None) == -1                # N % 64 != 0
    assert lib.svd_rowproj320(a16, 320, a16, None, a16, 961, 128, 960, 1, None) == -1                # odd output row stride (dword stores)
    assert lib.svd_rowproj320(a16, 316, a16, None, a16, 960, 128, 960, 1, None) == -1                # ldx < 320 / not a multiple of 8
    assert lib.svd_rowproj320(None, 320, a16, None, a16, 960, 128, 960, 1, None) == -1
    assert lib.svd_rowgemm320_pack_bytes() == 200 * 1024
    assert lib.svd_rowgemm320(a16, 320, a16, None, None, 0, 0, None, 0, None, 0, 1, None, None, 1e-5, None, 0, 128, 1, None) == -1       # neither Y nor Yn
    assert lib.svd_rowgemm320(a16, 320, a16, None, a16, 320, 48, None, 0, a16, 320, 1, None, None, 1e-5, None, 0, 128, 1, None) == -1    # rows_per_vec % 32 != 0
    assert lib.svd_rowgemm320(a16, 320, a16, None, None, 0, 0, None, 0, a16, 320, 1, None, None, 1e-5, a16, 320, 128, 1, None) == -1     # Yn without gamma / beta
    ff = lambda **kw: lib.svd_ff_geglu_fused(a16, 320, a16, kw.get("c", 320), kw.get("h", 1280), a16, kw.get("R"), 320, None, 0, 0.0, kw.get("r32", 0), a16, 320,
                                             kw.get("o32", 0), 128, 1, kw.get("rv"), 320, kw.get("rpv", 0), kw.get("g"), kw.get("b"), 1e-5, None, 0, 0, kw.get("yn"), 320, None)
    assert ff(c=640) == -1                                                                           # the fused kernel exists for dim 320
    assert ff(rv=a16, rpv=48) == -1                                                                  # per-frame vector: rows_per_vec % 32
    assert ff(yn=a16, g=a16, b=a16, R=a16, r32=0, o32=1) == -1                                       # fused LayerNorm needs the fp32 residual ...
    assert ff(yn=a16, g=a16, b=a16, R=a16, r32=1, o32=0) == -1                                       # ... and the fp32 output
    assert ff(yn=a16, R=a16, r32=1, o32=1) == -1                                                     # ... and gamma / beta


def test_product_path_does_not_import_oracle():
    """The shipped package must never reach into oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "streamingt2v_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_header_is_plain_c(tmp_path):
    """include/svdhip.h is the drop-in boundary: it must compile as C99 and as C++ on its own (no HIP / torch types in the signatures)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    src = tmp_path / "t.c"
    src.write_text('#include "svdhip.h"\nint f(void) { struct svd_gemm_args a; (void)a; return SVD_OK; }\n')
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only"], ["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++"]):
        r = subprocess.run(cmd + ["-I", inc, str(src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_ctypes_signatures_match_header_prototypes():
    """Every prototype of include/svdhip.h against the argtypes the binding declares: same arity, and per parameter the same class
    (pointer / stream -> c_void_p, int32_t -> c_int32, int64_t -> c_int64, float -> c_float, int* -> POINTER(c_int)).  A mismatch here
    would not crash -- it would silently pass garbage to a kernel."""
    from streamingt2v_amd.lib import lib
    txt = open(os.path.join(ROOT, "include", "svdhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = re.findall(r"\b(?:int|int64_t|const char\s*\*)\s*(svd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S)
    assert len(protos) >= 30
    checked = 0
    for name, params in protos:
        fn = getattr(lib, name)
        params = " ".join(params.split())
        plist = [] if params in ("void", "") else [p.strip() for p in params.split(",")]
        if fn.argtypes is None:
            assert not plist, f"{name}: no argtypes declared for {len(plist)} parameters"
            continue
        assert len(fn.argtypes) == len(plist), f"{name}: header has {len(plist)} parameters, binding {len(fn.argtypes)}"
        for i, (p, t) in enumerate(zip(plist, fn.argtypes)):
            if "*" in p:
                want = "pointer"
            elif p.startswith("svd_stream_t"):
                want = "pointer"
            elif p.startswith("int64_t"):
                want = ctypes.c_int64
            elif p.startswith("int32_t") or p.startswith("int "):
                want = ctypes.c_int32
            elif p.startswith("float"):
                want = ctypes.c_float
            elif p.startswith("double"):
                want = ctypes.c_double
            else:
                raise AssertionError(f"{name}: unhandled parameter type in '{p}'")
            if want == "pointer":
                assert t is ctypes.c_void_p or hasattr(t, "contents") or issubclass(t, ctypes._Pointer), f"{name} arg {i} ('{p}'): {t}"
            else:
                assert t is want or (want is ctypes.c_int32 and t is ctypes.c_int), f"{name} arg {i} ('{p}'): binding {t}, header wants {want}"
            checked += 1
    assert checked >= 250
