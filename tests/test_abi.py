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
