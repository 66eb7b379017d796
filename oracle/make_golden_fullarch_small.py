"""Shipped-architecture pins of the smaller networks (build container only):

    python oracle/make_golden_fullarch_small.py

  * temporal VideoDecoder (config.yaml:241-258: ch 128, ch_mult 1-2-4-4, 2 res blocks) on a 3-frame 8x8 latent  -> tests/golden/vae_fullarch.pt
  * sgm Encoder of the same size (the cond-frame / enhancer encoders) on one 64x64 image                       -> tests/golden/vae_enc_fullarch.pt
  * vendored EMA-VFI at F = 32 (the network vfi_init builds) on a 64x96 frame pair, fast TTA                   -> tests/golden/vfi_fullarch.pt
Each is the UNMODIFIED reference module with by-name weights; the oracle must agree (decoder / encoder <= 5e-4, EMA-VFI bit-exact).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap, vfi_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle import svd_oracle as O, vfi_oracle as OV  # noqa: E402
from oracle.cases import fullarch_small_inputs, vfi_weights  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_by_name(module, seed):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    sd = init_by_name(s, seed=seed)
    module.load_state_dict(sd, strict=True)
    return sd, s


def main():
    torch.set_grad_enabled(False)
    from models.svd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from models.svd.sgm.modules.diffusionmodules.model import Encoder
    inp = fullarch_small_inputs()
    kw = dict(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256,
              z_channels=4, double_z=True, attn_type="vanilla")
    dec = VideoDecoder(video_kernel_size=[3, 1, 1], **kw).eval()
    sd_d, _ = load_by_name(dec, seed=35)
    ref = dec(inp["z"], timesteps=inp["z"].shape[0])
    e = (ref - O.video_decoder(sd_d, O.VaeCfg(), inp["z"], inp["z"].shape[0])).abs().max().item()
    print(f"[decoder, shipped size] reference-vs-oracle {e:.3e} (|out| std {ref.std():.3f})")
    assert e <= 5e-4
    torch.save({"out": ref.clone()}, os.path.join(OUT, "vae_fullarch.pt"))
    enc = Encoder(**kw).eval()
    sd_e, _ = load_by_name(enc, seed=36)
    ref = enc(inp["x_enc"])
    e = (ref - O.vae_encoder(sd_e, O.VaeCfg(), inp["x_enc"])).abs().max().item()
    print(f"[encoder, shipped size] reference-vs-oracle {e:.3e} (|out| std {ref.std():.3f})")
    assert e <= 5e-4
    torch.save({"out": ref.clone()}, os.path.join(OUT, "vae_enc_fullarch.pt"))
    model = vfi_bootstrap.build_reference(F=32, depth=(2, 2, 2, 4, 4))
    spec = Spec()
    for k, v in model.net.state_dict().items():
        spec.add(k, *v.shape)
    sd_v = vfi_weights(spec, seed=12)
    model.net.load_state_dict(sd_v, strict=True)
    with vfi_bootstrap.cpu_only():
        tta = model.inference(inp["img0"], inp["img1"], TTA=True, fast_TTA=True)
    e = (tta - OV.inference_fast_tta(sd_v, OV.vfi_config(32, (2, 2, 2, 4, 4)), inp["img0"], inp["img1"])).abs().max().item()
    print(f"[EMA-VFI F=32] vendored-vs-oracle {e:.3e}")
    assert e <= 1e-5
    torch.save({"tta": tta.clone()}, os.path.join(OUT, "vfi_fullarch.pt"))
    print("wrote vae_fullarch.pt, vae_enc_fullarch.pt, vfi_fullarch.pt:", [os.path.getsize(os.path.join(OUT, f)) for f in ("vae_fullarch.pt", "vae_enc_fullarch.pt", "vfi_fullarch.pt")])


if __name__ == "__main__":
    main()
