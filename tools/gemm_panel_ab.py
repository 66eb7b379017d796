"""A/B of the GEMM's W-panel tile walk (csrc/gemm_impl.inc, round 6) on the job's plain GEMMs whose weight matrix exceeds an XCD's L2:
time per launch with SVD_GEMM_PANEL=0 (N fastest over the whole width) against the default (traffic model), same process, bit-compared.
    python tools/gemm_panel_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402

dt = torch.float16
ops.set_element_dtype(dt)
g = torch.Generator(device="cuda")
g.manual_seed(0)
# (name, M, N, K, geglu, fp32 residual / output)
CASES = [("stage 1 level-0 temporal q|k|v", 460800, 960, 320, False, False), ("stage 1 level-0 q|k", 460800, 640, 320, False, False),
         ("stage 1 level-1 temporal q|k|v", 115200, 1920, 640, False, False), ("stage 1 level-1 q|k", 115200, 1280, 640, False, False),
         ("enhancer level-0 q|k", 1094400, 640, 320, False, False),
         ("stage 1 level-1 GEGLU proj", 115200, 5120, 640, True, False), ("stage 1 level-2 GEGLU proj", 28800, 10240, 1280, True, False),
         ("stage 1 level-3 GEGLU proj", 7200, 10240, 1280, True, False), ("stage 1 level-1 ff down", 115200, 640, 2560, False, True),
         ("stage 1 level-2 ff down", 28800, 1280, 5120, False, True), ("stage 1 level-2 q|k", 28800, 2560, 1280, False, False),
         ("stage 1 level-2 temporal q|k|v", 28800, 3840, 1280, False, False), ("enhancer level-1 GEGLU proj", 273600, 5120, 640, True, False),
         ("enhancer level-2 GEGLU proj", 68400, 10240, 1280, True, False), ("enhancer level-2 ff down", 68400, 1280, 5120, False, True)]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'case':34s} {'M':>7s} {'N':>6s} {'K':>5s}   N-fastest      default walk   forced widths (us)")
for name, M, N, K, geglu, st in (CASES[:5] if "--small-w" in sys.argv else CASES):
    a = torch.randn(M, K, generator=g, device="cuda").to(dt)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn(M, N, generator=g, device="cuda") if st else None

    def run():
        return ops.gemm(a, w, bias=bias, geglu=geglu, residual=res, out_f32=st)
    out, ts = {}, {}
    for rep in range(3):                                  # interleaved, best of three: the clocks drift with the order of measurement
        for mode in ("0", None, "1", "2", "4", "8", "16"):
            if mode is None:
                os.environ.pop("SVD_GEMM_PANEL", None)
            else:
                os.environ["SVD_GEMM_PANEL"] = mode
            t = timed(run, reps=6)
            ts[mode] = min(ts.get(mode, 1e30), t)
            if rep == 0:
                out[mode] = run().clone()
    os.environ.pop("SVD_GEMM_PANEL", None)
    same = all(torch.equal(out["0"], v) for v in out.values())
    fl = 2.0 * M * N * K
    print(f"{name:34s} {M:7d} {N:6d} {K:5d} {ts['0']:8.1f} us {fl / ts['0'] / 1e6:6.0f} TF {ts[None]:8.1f} us {fl / ts[None] / 1e6:6.0f} TF   "
          + " ".join(f"{k}:{ts[k]:.0f}" for k in ("1", "2", "4", "8", "16")) + f"   bit-identical {same}", flush=True)
