"""svd_rowproj320 (csrc/rowproj.hip) against svd_gemm on the q | k (N = 640) and q | k | v (N = 960) projections of the 320-channel blocks, at the job's sizes:
time per launch (interleaved, best of three passes) and algorithmic bandwidth (X read once + Y written once).   python tools/rowproj_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402
from streamingt2v_amd.video_model import pack_rowproj320  # noqa: E402

dt = torch.float16
ops.set_element_dtype(dt)
torch.manual_seed(0)


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


print(f"{'M':>8s} {'N':>5s}   rowproj320          svd_gemm            ratio")
for M in (460800, 129024, 1094400):
    x = torch.randn(M, 320, device="cuda").to(dt)
    for N in (640, 960):
        w = torch.randn(N, 320) * 320 ** -0.5
        img, wd = pack_rowproj320(w).cuda(), w.to(dt).cuda()
        ta = tb = 1e30
        for _ in range(3):
            ta = min(ta, timed(lambda: ops.rowproj320(x, img, N)))
            tb = min(tb, timed(lambda: ops.gemm(x, wd)))
        nb = M * (320 + N) * 2.0
        print(f"{M:8d} {N:5d} {ta:8.1f} us {nb / ta / 1e6:5.2f} TB/s {tb:8.1f} us {nb / tb / 1e6:5.2f} TB/s   {tb / ta:5.2f}x", flush=True)
