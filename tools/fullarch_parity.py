"""Parity at the SHIPPED WIDTH: the full-architecture StreamingWrapper (VideoUNet 1.59 B + ControlNet 0.67 B parameters: 4 levels, 320..1280
channels, attention at every level, 2 res blocks) on a small latent against the REFERENCE's own output, tests/golden/wrapper_fullarch.pt
(oracle/make_golden_fullarch.py ran the unmodified reference modules on CPU; the oracle agrees with them to 4.9e-6 there).

    python tools/fullarch_parity.py [--dtype bf16|fp16|both] [--which svd|i2v|vae|both]

`--which vae` (also part of `both`): temporal VideoDecoder and sgm Encoder at the shipped size against tests/golden/vae_fullarch.pt /
vae_enc_fullarch.pt (oracle/make_golden_fullarch_small.py).  The encoder's mid attention sees only 64 tokens here (8x8 latent) and the
AEAttnBlock path wants pixels % 64 == 0 -- satisfied, but smaller than any shape tested so far.

`--which i2v`: the enhancer's I2VGenXLUNet at its shipped configuration (1.42 B parameters) on a 9x16 latent against
tests/golden/i2v_fullarch.pt (oracle/make_golden_i2v_fullarch.py: the unmodified vendored module; oracle agreement 3.2e-6).

The committed GPU tests compare against the reference at the tiny configuration (2 levels, 1 res block) and check the full size
structurally; this closes the gap between the two.  Written after the round's GPU budget was spent: run it first next round, then promote
it to a -m gpu test (latent 16x16 gives 256 / 64 / 16 / 4 tokens per frame at the four levels -- smaller than any shape tested so far).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--which", default="both")
    a = ap.parse_args()
    from oracle.cases import FULLARCH_CASE as c, fullarch_inputs
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.set_grad_enabled(False)
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "wrapper_fullarch.pt"))
    cfg = UNetConfig()
    sd_u, sd_c = init_by_name(VideoUNet(cfg).spec(), seed=c["seed_unet"]), init_by_name(ControlNet(cfg).spec(), seed=c["seed_cn"])
    inp = {k: v.cuda() for k, v in fullarch_inputs().items()}
    T = c["T"]

    def report(name, out, ref):
        out, ref = out.float().cpu(), ref.float()
        e, r = (out - ref).flatten(1).pow(2).mean(1).sqrt(), ref.flatten(1).pow(2).mean(1).sqrt()
        corr = torch.corrcoef(torch.stack([out.flatten(), ref.flatten()]))[0, 1].item()
        print(f"[{name}] per-frame L2 abs max {e.max():.3e} mean {e.mean():.3e} | rel max {(e / r).max():.3e} | corr {corr:.6f}")

    if a.which in ("both", "vae"):
        from oracle.cases import fullarch_small_inputs
        from streamingt2v_amd.temporal_ae import Encoder, VideoDecoder
        si = fullarch_small_inputs()
        gdir = os.path.join(ROOT, "tests", "golden")
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            if a.dtype not in ("both", name):
                continue
            ops.set_element_dtype(dt)
            dec = VideoDecoder()
            dec.load_state_dict(init_by_name(dec.spec(), seed=35), device="cuda")
            report(f"shipped-size VideoDecoder vs reference, {name}", dec.forward(si["z"].cuda(), timesteps=3), torch.load(os.path.join(gdir, "vae_fullarch.pt"))["out"])
            enc = Encoder()
            enc.load_state_dict(init_by_name(enc.spec(), seed=36), device="cuda")
            report(f"shipped-size VAE Encoder vs reference, {name}", enc(si["x_enc"].cuda()), torch.load(os.path.join(gdir, "vae_enc_fullarch.pt"))["out"])
            del dec, enc
            torch.cuda.empty_cache()
    if a.which in ("both", "i2v"):
        from oracle.cases import I2V_FULLARCH_CASE as ci, i2v_fullarch_inputs
        from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
        g2 = torch.load(os.path.join(ROOT, "tests", "golden", "i2v_fullarch.pt"))["out"]
        sd_e = init_by_name(I2VGenXLUNet(I2VConfig()).spec(), seed=ci["seed"])
        ei = i2v_fullarch_inputs()
        fr = lambda x: x.permute(0, 2, 1, 3, 4).reshape(-1, *x.shape[1:2], *x.shape[3:])
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            if a.dtype not in ("both", name):
                continue
            ops.set_element_dtype(dt)
            eu = I2VGenXLUNet(I2VConfig())
            eu.load_state_dict(sd_e, device="cuda")
            out = eu(ei["sample"], ei["t"], fps=ei["fps"], image_latents=ei["image_latents"], image_embeddings=ei["image_embeddings"],
                     encoder_hidden_states=ei["text"])[0]
            report(f"full-architecture I2VGenXLUNet vs vendored reference, {name}", fr(out), fr(g2))
            del eu
            torch.cuda.empty_cache()
        del sd_e
    if a.which not in ("both", "svd"):
        return
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        if a.dtype not in ("both", name):
            continue
        ops.set_element_dtype(dt)
        unet, cn = VideoUNet(cfg), ControlNet(cfg)
        unet.load_state_dict(sd_u, device="cuda")
        cn.load_state_dict(sd_c, device="cuda")
        wrap = StreamingWrapper(unet, cn, c["Tc"])
        out = wrap.forward(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")}, batch_size=2, num_video_frames=T,
                           image_only_indicator=torch.zeros(2, T, device="cuda"), ctrl_frames=inp["ctrl_frames"])
        report(f"full-architecture StreamingWrapper vs reference, {name}", out, gold["out"])
        out = unet.forward(torch.cat((inp["x"], inp["concat"]), 1), inp["t"], context=inp["crossattn"], y=inp["vector"], num_video_frames=T,
                           image_only_indicator=torch.zeros(2, T, device="cuda"))
        report(f"full-architecture VideoUNet (no control) vs reference, {name}", out, gold["out_noctrl"])
        del unet, cn, wrap
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
