"""Decoder precision plan sweep on the MI355X (round 6): per-frame L2 of the temporal VideoDecoder against the reference's fp32 output (tests/golden/vae_fullsize.pt,
2 frames -> 576x1024) and the time of one 8-frame decode group, for: all 16 bit | rim only | rim + fp32 stream at >= 512 / 256 / 128 channels.

    python tools/decoder_plan_sweep.py  -> profiles/r06_decoder_precision_plans.txt"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402
from streamingt2v_amd.params import init_by_name  # noqa: E402
from streamingt2v_amd.temporal_ae import VideoDecoder  # noqa: E402
from tools.fullsize_parity import decoder_fullsize  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    z8 = torch.randn(8, 4, 72, 128, generator=torch.Generator().manual_seed(1)).cuda()
    for rim, mc in ((False, 0), (True, 0), (True, 512), (True, 256), (True, 128)):
        ops.set_ae_precision_plan(rim, mc)
        r = decoder_fullsize("fp16")
        dec = VideoDecoder()
        dec.load_state_dict(init_by_name(dec.spec(), seed=35), device="cuda")
        for _ in range(2):
            dec.forward(z8, timesteps=8, clamp=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            dec.forward(z8, timesteps=8, clamp=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"[VideoDecoder fp16, exact rim {'on ' if rim else 'off'}, fp32 stream at >= {mc or 'inf':>3} channels] per-frame L2 vs reference fp32 (2 frames @576x1024): max {r['abs_max']:.3e} "
              f"mean {r['abs_mean']:.3e} corr {r['corr']:.7f} | decode of one 8-frame group {ms:.1f} ms", flush=True)
        del dec
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
