"""Generate tests/golden/*.pt from the UNMODIFIED reference (run in the build container only).

    python oracle/make_golden.py            # needs /root/reference ; writes tests/golden/

What it does, per case: build the reference nn.Modules (imported via oracle/ref_bootstrap.py), load the by-name
deterministic weights (streamingt2v_amd.params.init_by_name -- this also proves our state_dict spec equals the
reference's keys/shapes: load_state_dict(strict=True)), run the reference forward on seeded inputs, run the
restatement in oracle/svd_oracle.py on the same inputs, REQUIRE agreement (<= 2e-4 max abs), and store the
inputs + reference outputs as small fixtures.  Weights are NOT stored: they are re-derived from names + seed.
Nothing under tests/ or on the GPU box needs /root/reference afterwards.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle import svd_oracle as O  # noqa: E402
from oracle.cases import (TINY_UNET, TINY_VAE, tiny_unet_kwargs, tiny_wrapper_inputs, tiny_vae_inputs,  # noqa: E402
                          tiny_sampler_inputs)
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TOL = 2e-4


def spec_of(module):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    return s


def load_by_name(module, seed):
    sd = init_by_name(spec_of(module), seed=seed)
    module.load_state_dict(sd, strict=True)
    return sd


def maxerr(a, b):
    return (a - b).abs().max().item()


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper

    # ---------------- UNet + ControlNet + CAM (StreamingWrapper.forward) ----------------
    t0 = time.time()
    unet = VideoUNet(**tiny_unet_kwargs()).eval()
    sd_u = load_by_name(unet, seed=1)
    cn = ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity",
                              frame_expansion="none", downsample_controlnet_cond=True,
                              use_image_encoder_normalization=True, use_controlnet_mask=False, condition_encoder="",
                              conditioning_embedding_out_channels=list(TINY_UNET["cond_embed"])).eval()
    sd_c = load_by_name(cn, seed=2)
    # our host-side specs must equal the reference's state dicts (keys and shapes)
    from streamingt2v_amd.video_model import ControlNet as OurCN, UNetConfig, VideoUNet as OurUNet
    ours = UNetConfig(num_res_blocks=TINY_UNET["num_res_blocks"], attention_resolutions=TINY_UNET["attention_resolutions"],
                      channel_mult=TINY_UNET["channel_mult"], conditioning_embedding_out_channels=TINY_UNET["cond_embed"])
    assert dict(OurUNet(ours).spec()) == dict(spec_of(unet)), "UNet spec != reference state_dict"
    assert dict(OurCN(ours).spec()) == dict(spec_of(cn)), "ControlNet spec != reference state_dict"
    wrap = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=TINY_UNET["Tc"])
    inp = tiny_wrapper_inputs()
    kw = dict(batch_size=2, num_video_frames=TINY_UNET["T"], image_only_indicator=torch.zeros(2, TINY_UNET["T"]),
              ctrl_frames=inp["ctrl_frames"])
    ref = wrap(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")}, **dict(kw))
    cfg = O.Cfg(num_res_blocks=TINY_UNET["num_res_blocks"], attention_resolutions=TINY_UNET["attention_resolutions"],
                channel_mult=TINY_UNET["channel_mult"], cond_embed_channels=TINY_UNET["cond_embed"])
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    ora = O.streaming_wrapper(sd_u, sd_c, cfg, inp["x"], inp["t"], c, 2, TINY_UNET["T"], TINY_UNET["Tc"], inp["ctrl_frames"])
    e = maxerr(ref, ora)
    print(f"[wrapper] ref-vs-oracle max abs err {e:.3e}  (|ref| max {ref.abs().max():.3f}, std {ref.std():.3f})  {time.time() - t0:.1f}s")
    assert e <= TOL, e
    # plain UNet without ControlNet/CAM (config C2 path: hs_control_* = None)
    xcat = torch.cat((inp["x"], inp["concat"]), 1)
    ref_nc = unet(xcat, inp["t"], context=inp["crossattn"], y=inp["vector"], num_video_frames=TINY_UNET["T"],
                  image_only_indicator=torch.zeros(2, TINY_UNET["T"]))
    ora_nc = O.video_unet(sd_u, cfg, xcat, inp["t"], inp["crossattn"], inp["vector"], TINY_UNET["T"])
    e = maxerr(ref_nc, ora_nc)
    print(f"[unet no-ctrl] ref-vs-oracle max abs err {e:.3e}")
    assert e <= TOL, e
    torch.save({"out": ref.clone(), "out_noctrl": ref_nc.clone()}, os.path.join(OUT, "wrapper_tiny.pt"))

    # ---------------- sampler (EulerEDMSampler + Denoiser + guider) over the tiny wrapper ----------------
    t0 = time.time()
    from models.diffusion.discretizer import AlignYourSteps
    from models.svd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from models.svd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    T = TINY_UNET["T"]
    sampler = EulerEDMSampler(
        s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, num_steps=2, verbose=False, device="cpu",
        discretization_config={"target": "models.diffusion.discretizer.AlignYourSteps", "params": {"sigma_max": 700.0}},
        guider_config={"target": "models.svd.sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": 3.0, "min_scale": 1.5, "num_frames": T}})
    den = Denoiser({"target": "models.svd.sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sin = tiny_sampler_inputs()
    add = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
    z_ref = sampler(lambda a, s, cc: den(wrap, a, s, cc, **dict(add)), sin["noise"].clone(), cond=sin["c"], uc=sin["uc"])
    net = lambda a, cn_, cc: O.streaming_wrapper(sd_u, sd_c, cfg, a, cn_, cc, 2, T, TINY_UNET["Tc"], inp["ctrl_frames"])
    z_ora = O.euler_edm_sample(net, sin["noise"].clone(), sin["c"], sin["uc"], 2, T)
    e = maxerr(z_ref, z_ora)
    print(f"[sampler 2 steps] ref-vs-oracle max abs err {e:.3e} (|z| std {z_ref.std():.3f})  {time.time() - t0:.1f}s")
    assert e <= 5 * TOL, e
    sig30 = AlignYourSteps(sigma_max=700.0)(30, device="cpu")
    sig4 = AlignYourSteps(sigma_max=700.0)(4, device="cpu")
    assert maxerr(sig30, O.ays_sigmas(30)) == 0.0 and maxerr(sig4, O.ays_sigmas(4)) == 0.0
    from models.svd.sgm.modules.diffusionmodules.discretizer import EDMDiscretization
    edm25 = EDMDiscretization(sigma_min=0.002, sigma_max=700.0, rho=7.0)(25, device="cpu")
    assert maxerr(edm25, O.edm_sigmas(25)) == 0.0
    torch.save({"z": z_ref.clone(), "sigmas30": sig30, "sigmas4": sig4, "edm25": edm25}, os.path.join(OUT, "sampler_tiny.pt"))

    # ---------------- temporal VAE decoder ----------------
    t0 = time.time()
    dec = VideoDecoder(ch=TINY_VAE["ch"], out_ch=3, ch_mult=TINY_VAE["ch_mult"], num_res_blocks=TINY_VAE["num_res_blocks"],
                       attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256, z_channels=4, double_z=True,
                       attn_type="vanilla", video_kernel_size=[3, 1, 1]).eval()
    sd_d = load_by_name(dec, seed=3)
    from streamingt2v_amd.temporal_ae import VaeConfig, VideoDecoder as OurDec
    assert dict(OurDec(VaeConfig(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"])).spec()) == dict(spec_of(dec)), \
        "VideoDecoder spec != reference state_dict"
    z = tiny_vae_inputs()["z"]
    ref_d = dec(z, timesteps=z.shape[0])
    ora_d = O.video_decoder(sd_d, O.VaeCfg(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"]), z, z.shape[0])
    e = maxerr(ref_d, ora_d)
    print(f"[vae] ref-vs-oracle max abs err {e:.3e} (|out| std {ref_d.std():.3f})  {time.time() - t0:.1f}s")
    assert e <= TOL, e
    torch.save({"out": ref_d.clone()}, os.path.join(OUT, "vae_tiny.pt"))

    # ---------------- VAE encoder (conditioner side) ----------------
    from models.svd.sgm.modules.diffusionmodules.model import Encoder
    from streamingt2v_amd.temporal_ae import Encoder as OurEnc
    enc = Encoder(ch=TINY_VAE["ch"], out_ch=3, ch_mult=TINY_VAE["ch_mult"], num_res_blocks=TINY_VAE["num_res_blocks"], attn_resolutions=[],
                  dropout=0.0, in_channels=3, resolution=256, z_channels=4, double_z=True, attn_type="vanilla").eval()
    sd_e = load_by_name(enc, seed=4)
    assert dict(OurEnc(VaeConfig(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"])).spec()) == dict(spec_of(enc)), \
        "Encoder spec != reference state_dict"
    xe = tiny_vae_inputs()["x_enc"]
    ref_e = enc(xe)
    ora_e = O.vae_encoder(sd_e, O.VaeCfg(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"]), xe)
    e = maxerr(ref_e, ora_e)
    print(f"[vae encoder] ref-vs-oracle max abs err {e:.3e} (|out| std {ref_e.std():.3f})")
    assert e <= TOL, e
    torch.save({"out": ref_e.clone()}, os.path.join(OUT, "vae_enc_tiny.pt"))
    with torch.device("meta"):
        fenc = Encoder(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3,
                       resolution=256, z_channels=4, double_z=True, attn_type="vanilla")
    assert dict(OurEnc(VaeConfig()).spec()) == dict(spec_of(fenc)), "full encoder spec mismatch"

    # cond_frames embedder's encoder: AutoencoderKLModeOnly.encode (encoder + quant_conv, mode of the posterior)
    from models.svd.sgm.models.autoencoder import AutoencoderKLModeOnly
    from streamingt2v_amd.temporal_ae import CondFrameEncoder
    kl = AutoencoderKLModeOnly(embed_dim=4, monitor="val/rec_loss", lossconfig={"target": "torch.nn.Identity"},
                               ddconfig=dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3,
                                             ch=TINY_VAE["ch"], ch_mult=list(TINY_VAE["ch_mult"]), num_res_blocks=TINY_VAE["num_res_blocks"],
                                             attn_resolutions=[], dropout=0.0)).eval()
    ours_kl = CondFrameEncoder(VaeConfig(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"]))
    sd_k = init_by_name(ours_kl.spec(), seed=6)
    missing, unexpected = kl.load_state_dict(sd_k, strict=False)
    assert not unexpected and all(k.startswith(("decoder.", "post_quant_conv.")) for k in missing), (missing[:4], unexpected[:4])
    ref_k = kl.encode(xe)
    ora_k = O.cond_frame_encode(sd_k, O.VaeCfg(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"]), xe)
    e = maxerr(ref_k, ora_k)
    print(f"[cond-frame encoder (KL mode only)] ref-vs-oracle max abs err {e:.3e} (|out| std {ref_k.std():.3f})")
    assert e <= TOL, e
    torch.save({"out": ref_k.clone()}, os.path.join(OUT, "cond_enc_tiny.pt"))

    # 2-D decoder (sgm Decoder == the arithmetic of diffusers' AutoencoderKL decoder around the enhancer)
    from models.svd.sgm.modules.diffusionmodules.model import Decoder
    from streamingt2v_amd.temporal_ae import Decoder2D
    d2 = Decoder(ch=TINY_VAE["ch"], out_ch=3, ch_mult=TINY_VAE["ch_mult"], num_res_blocks=TINY_VAE["num_res_blocks"], attn_resolutions=[],
                 dropout=0.0, in_channels=3, resolution=256, z_channels=4, attn_type="vanilla").eval()
    sd_2 = load_by_name(d2, seed=7)
    assert dict(Decoder2D(VaeConfig(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"])).spec()) == dict(spec_of(d2)), "Decoder2D spec"
    z2 = tiny_vae_inputs()["z"][:2]
    ref_2 = d2(z2)
    ora_2 = O.vae_decoder_2d(sd_2, O.VaeCfg(TINY_VAE["ch"], TINY_VAE["ch_mult"], TINY_VAE["num_res_blocks"]), z2)
    e = maxerr(ref_2, ora_2)
    print(f"[2-D decoder] ref-vs-oracle max abs err {e:.3e} (|out| std {ref_2.std():.3f})")
    assert e <= TOL, e
    torch.save({"out": ref_2.clone()}, os.path.join(OUT, "vae_dec2d_tiny.pt"))

    # full-size specs (meta device): key/shape equality of the shipped configuration
    with torch.device("meta"):
        from oracle.cases import full_unet_kwargs
        fu = VideoUNet(**full_unet_kwargs())
        fdec = VideoDecoder(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0,
                            in_channels=3, resolution=256, z_channels=4, double_z=True, attn_type="vanilla",
                            video_kernel_size=[3, 1, 1])
    assert dict(OurUNet(UNetConfig()).spec()) == dict(spec_of(fu)), "full UNet spec mismatch"
    assert dict(OurDec(VaeConfig()).spec()) == dict(spec_of(fdec)), "full decoder spec mismatch"
    print("full-size state_dict specs equal the reference (UNet %d tensors, decoder %d tensors)"
          % (len(spec_of(fu)), len(spec_of(fdec))))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
