"""Parity at the SHIPPED WIDTH: the full-architecture VideoUNet (all 1.59 B parameters: 4 levels, 320..1280 channels, transformer depth 1,
2 res blocks) on a small latent against the CPU oracle, which needs ~15-60 s for this size instead of ~7 min per full-size forward.

    python tools/fullarch_parity.py [--latent 32 32] [--frames 2] [--dtype bf16|fp16|both]

The committed GPU tests compare against the oracle at the tiny configuration (2 levels) and check the full size structurally; this closes
the gap between the two (not yet part of the -m gpu suite: written after the round's GPU budget was spent -- run it first next round).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, nargs=2, default=[32, 32])
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--dtype", default="both")
    a = ap.parse_args()
    from oracle import svd_oracle as O
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    torch.set_grad_enabled(False)
    T, (h, w) = a.frames, a.latent
    cfg = UNetConfig(controlnet_mode=False)
    sd = init_by_name(VideoUNet(cfg).spec(), seed=33)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2 * T, 8, h, w, generator=g)
    t = torch.randn(2 * T, generator=g) * 0.5
    ctx, vec = torch.randn(2 * T, 1, 1024, generator=g), torch.randn(2 * T, 768, generator=g) * 0.5
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref = O.video_unet(sd, O.Cfg(), x, t, ctx, vec, T)
    print(f"oracle: {time.time() - t0:.1f} s, output std {ref.std():.3f}")
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        if a.dtype not in ("both", name):
            continue
        ops.set_element_dtype(dt)
        unet = VideoUNet(cfg)
        unet.load_state_dict(sd, device="cuda")
        out = unet.forward(x.cuda(), t.cuda(), context=ctx.cuda(), y=vec.cuda(), num_video_frames=T,
                           image_only_indicator=torch.zeros(2, T, device="cuda")).float().cpu()
        e = (out - ref).flatten(1).pow(2).mean(1).sqrt()
        r = ref.flatten(1).pow(2).mean(1).sqrt()
        corr = torch.corrcoef(torch.stack([out.flatten(), ref.flatten()]))[0, 1].item()
        print(f"[full-architecture VideoUNet @{h}x{w}, {T} frames, {name}] per-frame L2 abs max {e.max():.3e} | rel max {(e / r).max():.3e} | corr {corr:.6f}")
        del unet
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
