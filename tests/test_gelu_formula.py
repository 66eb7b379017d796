"""CPU suite: the GEGLU epilogue's erf-GELU (svd_common.h gelu_erf_f: max(x,0) - |x| * 2^(r(min(|x|,7)) - 1), r = degree-6 polynomial without constant
term) against the exact 0.5 x (1 + erf(x / sqrt 2)) of the reference's GEGLU (attention.py:99-101).  The coefficients are read from the header and the
sequence is evaluated in float32 exactly as the kernel does (fma order included, up to the fused rounding of fmaf)."""
import os
import re

import numpy as np
from scipy.special import erf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernel_coefficients():
    src = open(os.path.join(ROOT, "streamingt2v_amd", "csrc", "svd_common.h")).read()
    body = src[src.index("__device__ __forceinline__ float gelu_erf_f(float x)"):]
    body = body[:body.index("\n}\n")]
    c = [float(v) for v in re.findall(r"(-?\d\.\d+e[+-]\d+)f", body)]
    assert len(c) == 6 and "fminf(fabsf(x), 7.0f)" in body and "fmaf(r, t, -1.0f)" in body, body
    return c          # c6, c5, c4, c3, c2, c1 in Horner order


def test_single_exp_gelu_matches_exact_erf_gelu():
    c = np.array(_kernel_coefficients(), dtype=np.float32)
    x = np.concatenate([np.linspace(-40, 40, 2_000_001), np.linspace(-1e-3, 1e-3, 2001), [0.0, -0.0, 7.0, -7.0, 1e4, -1e4]]).astype(np.float32)
    t = np.minimum(np.abs(x), np.float32(7.0))
    r = c[0] * t + c[1]
    for k in range(2, 6):
        r = r * t + c[k]
    q = np.exp2((r * t - np.float32(1.0)).astype(np.float32)).astype(np.float32)
    got = (np.maximum(x, np.float32(0)) - np.abs(x) * q).astype(np.float64)
    xd = x.astype(np.float64)
    ref = 0.5 * xd * (1.0 + erf(xd / np.sqrt(2.0)))
    err = np.abs(got - ref)
    small = np.abs(xd) <= 40
    assert err[small].max() <= 4e-6, err[small].max()                       # 1/50 of the 16-bit rounding of an O(1) activation
    assert (err / np.maximum(np.abs(ref), 1e-2)).max() <= 5e-5              # relative, away from the zero crossing
    assert got[x == 0].tolist() == [0.0] * int((x == 0).sum())
    assert abs(got[-2] - 1e4) <= 1e-3 and abs(got[-1]) <= 1e-6               # far tails: identity / zero
