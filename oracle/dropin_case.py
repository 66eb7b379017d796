"""The executed drop-in of INTEGRATION.md section 1 -- TEST INFRASTRUCTURE ONLY (build container: needs /root/reference).

`run(swap)` drives the reference's UNMODIFIED stage-1 chunk code on CPU:

    StreamingSVD._generate_conditional_output   code/diffusion_trainer/streaming_svd.py:155-221   (unbound, on a bare namespace)
      EulerEDMSampler (AlignYourSteps, LinearPredictionGuider)   sgm/modules/diffusionmodules/sampling.py:41-52,93-130
        Denoiser + VScalingWithEDMcNoise                         denoiser.py:23-39
          network(input * c_in, c_noise, cond, **additional_model_inputs)        <- inference_model
      StreamingSVD.decode_first_stage                            streaming_svd.py:123-151  (incl. the isinstance(..., VideoDecoder) at :138)
        AutoencodingEngine.decode -> self.decoder(z, timesteps=n) sgm/models/autoencoder.py:210-212

swap = False: inference_model / decoder are the reference's own StreamingWrapper(VideoUNet, ControlNet) / VideoDecoder.
swap = True : they are replaced EXACTLY as INTEGRATION.md section 1 prescribes -- our StreamingWrapper / VideoUNet / ControlNet / VideoDecoder
              loaded from the reference modules' state_dict, `model.inference_model = ...`, `model.first_stage_model.decoder = ...`, and the
              module-level name `VideoDecoder` of diffusion_trainer/streaming_svd.py rebound to ours (the import edit the isinstance needs).
              Our classes run on tests/svd_shim.py's fp32 torch statements of the HIP launchers (no GPU in the build container).
The conditioner is a real GeneralConditioner around linear stand-in embedders (as oracle/make_golden_conditioner.py).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STEPS = 3
CASE = dict(H=64, W=64, T=8, Tc=3, seed=4712, vae_ch=32, vae_ch_mult=(1, 1, 2, 2), vae_res=1)     # 4-level decoder: the reference code assumes 8x (:157)


def case_inputs():
    g = torch.Generator(); g.manual_seed(97)
    c = CASE
    return dict(frame=torch.rand(3, c["H"], c["W"], generator=g) * 2 - 1, ctrl_frames=torch.rand(1, c["Tc"], 3, c["H"], c["W"], generator=g) * 2 - 1)


def _conditioner(M):
    from oracle.cases import fake_clip_embed, fake_cond_encode

    class FakeClip(M.AbstractEmbModel):
        def forward(self, img):
            return fake_clip_embed(img)

    class FakeEncoder(nn.Module):
        def encode(self, x):
            return fake_cond_encode(x)

    fakes = types.ModuleType("oracle_ref_fakes")
    fakes.FakeClip, fakes.FakeEncoder = FakeClip, FakeEncoder
    sys.modules["oracle_ref_fakes"] = fakes
    P = "models.svd.sgm.modules.encoders.modules."
    return M.GeneralConditioner([
        dict(is_trainable=False, input_key="cond_frames_without_noise", target=P + "FrozenOpenCLIPImagePredictionEmbedder",
             params=dict(n_cond_frames=1, n_copies=1, open_clip_embedding_config=dict(target="oracle_ref_fakes.FakeClip", params={}))),
        dict(input_key="fps_id", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="motion_bucket_id", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="cond_frames", is_trainable=False, target=P + "VideoPredictionEmbedderWithEncoder",
             params=dict(disable_encoder_autocast=True, n_cond_frames=1, n_copies=1, is_ae=True, encoder_config=dict(target="oracle_ref_fakes.FakeEncoder", params={}))),
        dict(input_key="cond_aug", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
    ])


def run(swap, monkeypatch=None, record=None):
    """-> frames [T, 3, H, W] fp32 in [-1, 1] of one conditional chunk (STEPS Euler steps + decode + clamp).
    record: a list -> every call the reference's code makes on `inference_model` (Denoiser: network(input * c_in, c_noise, cond, **kw)) and on
    `first_stage_model.decoder` (AutoencodingEngine.decode: decoder(z, timesteps=n)) is appended as (kind, args, kwargs, output) -- the fixture
    tests/test_gpu_dropin.py replays through the HIP mirrors on the GPU box, where /root/reference does not exist."""
    from oracle import ar_bootstrap
    from oracle.cases import TINY_UNET, tiny_unet_kwargs
    c = CASE
    from streamingt2v_amd.params import Spec, init_by_name
    torch.set_grad_enabled(False)
    Ref = ar_bootstrap.install()
    ref_mod = importlib.import_module("diffusion_trainer.streaming_svd")
    M = importlib.import_module("models.svd.sgm.modules.encoders.modules")
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.models.autoencoder import AutoencodingEngine
    from models.svd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    from models.svd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from models.svd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    T, Tc = c["T"], c["Tc"]

    def by_name(module, seed):
        s = Spec()
        for k, v in module.state_dict().items():
            s.add(k, *v.shape)
        module.load_state_dict(init_by_name(s, seed=seed), strict=True)
        return module

    unet = by_name(VideoUNet(**tiny_unet_kwargs()).eval(), 1)
    cn = by_name(ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity", frame_expansion="none",
                                      downsample_controlnet_cond=True, use_image_encoder_normalization=True, use_controlnet_mask=False,
                                      condition_encoder="", conditioning_embedding_out_channels=list(TINY_UNET["cond_embed"])).eval(), 2)
    dec = by_name(VideoDecoder(ch=c["vae_ch"], out_ch=3, ch_mult=list(c["vae_ch_mult"]), num_res_blocks=c["vae_res"], attn_resolutions=[],
                               dropout=0.0, in_channels=3, resolution=256, z_channels=4, double_z=True, attn_type="vanilla",
                               video_kernel_size=[3, 1, 1]).eval(), 3)
    first_stage = AutoencodingEngine.__new__(AutoencodingEngine)          # decode() is all the path uses: `return self.decoder(z, **kwargs)`
    nn.Module.__init__(first_stage)
    first_stage.decoder = dec
    sampler = EulerEDMSampler(
        s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, num_steps=STEPS, verbose=False, device="cpu",
        discretization_config={"target": "models.diffusion.discretizer.AlignYourSteps", "params": {"sigma_max": 700.0}},
        guider_config={"target": "models.svd.sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                       "params": {"max_scale": 3.0, "min_scale": 1.5, "num_frames": T}})
    class BareStreamingSVD(nn.Module):
        """Stand-in for the LightningModule shell of the reference's StreamingSVD: an nn.Module whose hot-path objects are REGISTERED
        children, exactly like the real one (a plain object cannot be assigned over them), and whose methods are the reference's own."""
    BareStreamingSVD.__module__ = ref_mod.__name__                      # dropin.install finds the isinstance'd name through the model's module
    model = BareStreamingSVD()
    model.sampler = sampler
    model.conditioner = _conditioner(M)
    model.use_memopt = False
    model.denoiser = Denoiser({"target": "models.svd.sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    model.model = OpenAIWrapper(unet)                                   # state_dict keys model.diffusion_model.*, like the checkpoint
    model.controlnet = cn
    model.inference_model = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=Tc)
    model.first_stage_model = first_stage
    model.inference_params = types.SimpleNamespace(num_conditional_frames=Tc)
    model.diff_trainer_params = types.SimpleNamespace(scale_factor=0.18215, disable_first_stage_autocast=True)
    object.__setattr__(model, "device", "cpu")
    for name in ("get_batch_sgm", "get_unique_embedder_keys_from_conditioner", "decode_first_stage"):
        object.__setattr__(model, name, types.MethodType(getattr(Ref, name), model))

    if record is not None:
        cl = lambda v: (v.detach().clone() if torch.is_tensor(v) else ({k: cl(x) for k, x in v.items()} if isinstance(v, dict) else v))
        model.inference_model.register_forward_hook(
            lambda m, args, kwargs, out: record.append(("network", [cl(x) for x in args], {k: cl(x) for k, x in kwargs.items()}, cl(out))), with_kwargs=True)
        dec.register_forward_hook(
            lambda m, args, kwargs, out: record.append(("decoder", [cl(x) for x in args], {k: cl(x) for k, x in kwargs.items()}, cl(out))), with_kwargs=True)
        record.append(("sigmas", [sampler.discretization(STEPS, device="cpu").double().clone()], {}, None))
        record.append(("guider_scale", [sampler.guider.scale.clone()], {}, None))
    # what the reference's `on_inference_epoch_start` does at the start of EVERY predict epoch (streaming_svd.py:46-55; its `super()` call needs the Lightning
    # class, so the two statements are restated on the bare shell): it rebuilds the wrapper from the reference's own networks
    def on_inference_epoch_start():
        model.inference_model = StreamingWrapper(diffusion_model=model.model.diffusion_model, controlnet=model.controlnet,
                                                 num_frame_conditioning=model.inference_params.num_conditional_frames)
        model.inference_model.requires_grad_(False)
    object.__setattr__(model, "on_inference_epoch_start", on_inference_epoch_start)
    undo = None
    if swap:
        # ------------------------------------------------------------------ INTEGRATION.md section 1, verbatim in substance --------------
        from tests import svd_shim
        svd_shim.install(monkeypatch)                                                     # (CPU stand-in for libsvdhip.so's launchers)
        from streamingt2v_amd import dropin
        from streamingt2v_amd.temporal_ae import VaeConfig
        from streamingt2v_amd.video_model import UNetConfig
        ucfg = UNetConfig(num_res_blocks=TINY_UNET["num_res_blocks"], attention_resolutions=TINY_UNET["attention_resolutions"],
                          channel_mult=TINY_UNET["channel_mult"], conditioning_embedding_out_channels=TINY_UNET["cond_embed"])
        prev = ref_mod.VideoDecoder
        dropin.install(model, device="cpu", unet_cfg=ucfg, vae_cfg=VaeConfig(c["vae_ch"], c["vae_ch_mult"], c["vae_res"]))     # <- THE swap
        assert type(model.inference_model).__name__ == "HipModule" and ref_mod.VideoDecoder is dropin.VideoDecoderModule
        model.on_inference_epoch_start()                  # trainer.predict fires it AFTER an install at the end of init_model: the swap must survive it
        assert type(model.inference_model).__name__ == "HipModule", "on_inference_epoch_start put the reference's own wrapper back"
        undo = lambda: setattr(ref_mod, "VideoDecoder", prev)
        # ----------------------------------------------------------------------------------------------------------------------------------
    try:
        inp = case_inputs()
        torch.manual_seed(c["seed"])                      # cond-aug rand_like and the sampler noise come from the global stream (:174, :203)
        out = Ref._generate_conditional_output(model, inp["frame"], model.inference_params, ctrl_frames=inp["ctrl_frames"])
    finally:
        if undo:
            undo()
    assert out.shape == (T, 3, c["H"], c["W"]) and float(out.min()) >= -1 and float(out.max()) <= 1
    return out
