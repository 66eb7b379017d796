"""HBM bandwidth of the norm / element-wise kernels at the shapes of the job -> profiles/r03_norm_bandwidth.txt (round-2 verdict: "no
profiles/ file gives GB/s for gn_*, layernorm_kernel, copy_rows, ae_time_mix3").

    python tools/norm_bench.py > gpurun_out/r03_norm_bandwidth.txt

For every shape: time of the launcher (HIP events, best of 5 after 2 warm-ups) and ALGORITHMIC bytes / time = every operand moved once
(GroupNorm = statistics pass reads x, apply pass reads x and writes y; LayerNorm reads x, writes y).  Shapes: the UNet levels of an AR
forward (CFG 2 x 25 frames at latent 72x128), the ControlNet's 14 frames, the temporal VAE's levels for one 8-frame group (up to 128
channels at 576x1024).  Inputs: 16-bit, and fp32 (the fp32 residual stream of round 3).  Peak: 8.0 TB/s HBM3E."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402


def t(fn, n=5):
    for _ in range(2):
        fn()
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


def line(name, shape, ms, nbytes):
    print(f"{name:34s} {shape:26s} {ms * 1e3:9.1f} us {nbytes / 1e6:9.1f} MB {nbytes / ms / 1e9:7.2f} TB/s  {nbytes / ms / 1e9 / 8.0:6.3f} of 8 TB/s", flush=True)


def main():
    ops.set_element_dtype(torch.float16)
    print(f"{'kernel(s)':34s} {'frames x pixels x C (input)':26s} {'time':>12s} {'alg. bytes':>12s} {'bandwidth':>12s}")
    unet = ((50, 9216, 320), (50, 2304, 640), (50, 576, 1280), (50, 144, 1280), (14, 9216, 320), (50, 9216, 640), (50, 2304, 1280))
    vae = ((8, 9216, 512), (8, 36864, 512), (8, 147456, 256), (8, 589824, 128))
    for frames, pix, C in unet + vae:
        for dt in (torch.float16, torch.float32):
            if dt == torch.float32 and (frames, pix, C) in vae:
                continue                                  # the VAE keeps a 16-bit stream
            x = torch.randn(frames * pix, C, device="cuda").to(dt)
            g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            es, shape = x.element_size(), f"{frames} x {pix} x {C} ({'fp32' if dt == torch.float32 else 'fp16'})"
            n = x.numel()
            ms = t(lambda: ops.groupnorm(x, frames, pix, g, b, 1e-5, silu=True))
            line("gn_stats_partial+finalize+apply", shape, ms, n * (2 * es + 2))
            if (frames, pix, C) in unet and C <= 2048:
                ms = t(lambda: ops.layernorm(x, g, b))
                line("layernorm", shape, ms, n * (es + 2))
            if dt == torch.float32:
                ms = t(lambda: ops.to_elem_rows(x))
                line("cast_rows_f32", shape, ms, n * 6)
                y = torch.randn(frames * pix, C, device="cuda").to(torch.float16)
                ms = t(lambda: ops.add_rows(x, y))
                line("add_rows_f32", shape, ms, n * 10)
            del x
    a, b2 = (torch.randn(50 * 9216, 320, device="cuda").half() for _ in range(2))
    ms = t(lambda: ops.concat_channels(a, b2))
    line("copy_rows x2 (concat)", "50 x 9216 x 320+320 (fp16)", ms, a.numel() * 2 * 4)
    af, bf = a.float(), b2.float()
    ms = t(lambda: ops.concat_channels(af, bf))
    line("cast_rows_f32 x2 (concat)", "50 x 9216 x 320+320 (fp32)", ms, a.numel() * 2 * 6)
    # AE3DConv.time_mix_conv at the decoder's output: reads the 4-channel fp32 token rows of 3 frames per output frame (L2-resident), writes NCHW fp32
    F_, H, W = 8, 576, 1024
    xt = torch.randn(F_ * H * W, 4, device="cuda")
    w, bb = torch.randn(3, 3, 3, device="cuda"), torch.zeros(3, device="cuda")
    ms = t(lambda: ops.ae_time_mix3(xt, w, bb, F_, H, W, True))
    line("ae_time_mix3", f"{F_} x {H * W} x 4 -> NCHW 3 (fp32)", ms, F_ * H * W * (16 + 12))
    x0 = torch.randn(50, 4, 72, 128, device="cuda")
    ms = t(lambda: ops.tokens_to_nchw(ops.nchw_to_tokens(x0, x0, None, 32), 4, 50, 72, 128))
    line("nchw_to_tokens + tokens_to_nchw", "50 x 9216 x (4+4 -> 32 -> 4)", ms, 50 * 9216 * (32 + 64 + 64 + 16))


if __name__ == "__main__":
    main()
