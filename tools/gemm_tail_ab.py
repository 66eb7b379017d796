"""A/B of the GEMM tail split (csrc/gemm_impl.inc launch_kernel; SVD_GEMM_TAIL=0|1) on the level-0 launches of the stage-1 job whose last round is nearly empty
(1 800 tiles of 256 rows on 256 CUs).  One process, interleaved, best of three; bit-compared.     python tools/gemm_tail_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402
from streamingt2v_amd.video_model import pack_conv3x3, pack_tconv3  # noqa: E402

dt = torch.float16
ops.set_element_dtype(dt)
torch.manual_seed(0)


def timed(fn, reps=8):
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


Fr, H, W = 50, 72, 128
M = Fr * H * W
cases = []
x320 = torch.randn(M, 320, device="cuda").to(dt)
x640 = torch.randn(M, 640, device="cuda").to(dt)
r32 = torch.randn(M, 320, device="cuda")
wc = pack_conv3x3(torch.randn(320, 320, 3, 3) * (9 * 320) ** -0.5).to(dt).cuda()
wc2 = pack_conv3x3(torch.randn(320, 640, 3, 3) * (9 * 640) ** -0.5).to(dt).cuda()
wt = pack_tconv3(torch.randn(320, 320, 3, 1, 1) * (3 * 320) ** -0.5).to(dt).cuda()
wl = (torch.randn(320, 640) * 640 ** -0.5).to(dt).cuda()
b = torch.randn(320, device="cuda")
cases.append(("conv3x3 320 -> 320 (+ fp32 residual)", lambda: ops.gemm(x320, wc, bias=b, residual=r32, out_f32=True, conv=dict(cin=320, hin=H, win=W, hout=H, wout=W, stride=1, ups=0, frames=Fr)), 2.0 * M * 320 * 2880))
cases.append(("conv3x3 640 -> 320 (decoder skip)", lambda: ops.gemm(x640, wc2, bias=b, conv=dict(cin=640, hin=H, win=W, hout=H, wout=W, stride=1, ups=0, frames=Fr)), 2.0 * M * 320 * 5760))
cases.append(("temporal 3-tap 320 -> 320", lambda: ops.gemm(x320, wt, bias=b, residual=r32, out_f32=True, temporal=dict(cin=320, T=25, pix=H * W)), 2.0 * M * 320 * 960))
cases.append(("plain 640 -> 320 (skip 1x1)", lambda: ops.gemm(x640, wl, bias=b, out_f32=True), 2.0 * M * 320 * 640))
print(f"{'launch (M = 460 800)':40s}   unsplit            tail split")
for name, fn, fl in cases:
    best, outs = {}, {}
    for rep in range(3):
        for v in ("0", "1"):
            os.environ["SVD_GEMM_TAIL"] = v
            best[v] = min(best.get(v, 1e30), timed(fn))
            if rep == 0:
                outs[v] = fn().clone()
    os.environ.pop("SVD_GEMM_TAIL", None)
    print(f"{name:40s} {best['0']:8.1f} us {fl / best['0'] / 1e6:5.0f} TF {best['1']:8.1f} us {fl / best['1'] / 1e6:5.0f} TF   bit-identical {torch.equal(outs['0'], outs['1'])}", flush=True)
