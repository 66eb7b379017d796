"""The reference-side swap of INTEGRATION.md section 1 as code: put the MI355X hot path inside the reference's own StreamingSVD object.

The reference's `StreamingSVD` (code/diffusion_trainer/streaming_svd.py) and `AutoencodingEngine` (sgm/models/autoencoder.py) are
`torch.nn.Module`s and `inference_model` / `first_stage_model.decoder` are REGISTERED child modules: assigning a plain object to them raises
`TypeError: cannot assign ... as child module` (found by executing the swap, tests/test_dropin_reference.py).  The mirrors in this package are
plain Python objects (they own packed device tensors, not nn.Parameters), so the swap goes through `HipModule`, a parameter-less nn.Module
that forwards `forward(...)` and attribute reads to the object it wraps:

    from streamingt2v_amd import dropin
    dropin.install(self.model)            # end of inference_i2v.StreamingPipeline.init_model(), after the strict checkpoint load (:125-141)

After it, the reference's unmodified `_generate_conditional_output` / `Denoiser` / `EulerEDMSampler` / `decode_first_stage` drive
`StreamingWrapper.forward(x * c_in, c_noise, cond, batch_size=, num_video_frames=, image_only_indicator=, ctrl_frames=)`
(denoiser.py:36-38, wrappers.py:23-78) and `decoder(z, timesteps=n)` (autoencoder.py:210-212) on libsvdhip.so.
"""
import sys

import torch.nn as nn


class HipModule(nn.Module):
    """nn.Module shell around one of this package's objects (StreamingWrapper, VideoDecoder, I2VGenXLUNet): no parameters, `forward` and
    every other attribute come from the wrapped object."""

    def __init__(self, impl):
        super().__init__()
        object.__setattr__(self, "impl", impl)

    def forward(self, *args, **kwargs):
        return self.impl.forward(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.__dict__["impl"], name)


class VideoDecoderModule(HipModule):
    """The class `decode_first_stage` tests with isinstance(self.first_stage_model.decoder, VideoDecoder) (streaming_svd.py:138) to decide
    whether to pass `timesteps=`: `install` rebinds the module-level name `VideoDecoder` of the reference's streaming_svd.py to this class."""


def install(model, device="cuda", unet_cfg=None, vae_cfg=None, state_dict=None, offload_reference=False):
    """model: the reference's StreamingSVD module AFTER its strict checkpoint load.  Builds the MI355X mirrors from the same weights
    (`model.diffusion_model.*`, `controlnet.*`, `first_stage_model.decoder.*`; strict) and swaps the two hot-path objects in place.
    Returns (wrapper, decoder): the wrapped streamingt2v_amd objects.

    The reference's own `model.model` (the UNet wrapper) and `model.controlnet` stay REGISTERED children of `model` after the swap: their
    weights are not used by the hot path any more but stay wherever they were -- on a GPU that is ~9 GB fp32 next to the packed 16-bit copies,
    and `model.state_dict()` / `.to()` / `.half()` keep operating on them (the mirrors do not follow such calls: re-run install() after changing
    weights).  offload_reference=True moves those two modules (and nothing else) to the CPU and empties the allocator cache; the state dict
    stays complete."""
    from .temporal_ae import VaeConfig, VideoDecoder
    from .video_model import ControlNet, UNetConfig, VideoUNet
    from .wrappers import StreamingWrapper
    sd = state_dict if state_dict is not None else model.state_dict()
    unet = VideoUNet(unet_cfg or UNetConfig()).load_state_dict(sd, device=device, prefix="model.diffusion_model.")
    cnet = ControlNet.from_unet(unet).load_state_dict(sd, device=device, prefix="controlnet.")
    dec = VideoDecoder(vae_cfg or VaeConfig()).load_state_dict(sd, device=device, prefix="first_stage_model.decoder.")
    wrapper = StreamingWrapper(diffusion_model=unet, controlnet=cnet, num_frame_conditioning=model.inference_params.num_conditional_frames)
    model.inference_model = HipModule(wrapper)
    model.first_stage_model.decoder = VideoDecoderModule(dec)       # AutoencodingEngine.decode() keeps calling self.decoder(z, timesteps=n)
    ref_module = sys.modules.get(type(model).__module__)
    if ref_module is not None and hasattr(ref_module, "VideoDecoder"):
        ref_module.VideoDecoder = VideoDecoderModule                 # the isinstance at streaming_svd.py:138
    if offload_reference:
        import torch
        for name in ("model", "controlnet"):
            m = getattr(model, name, None)
            if isinstance(m, nn.Module):
                m.to("cpu")
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    return wrapper, dec


def install_enhancer(pipeline, device="cuda", cfg=None, state_dict=None):
    """pipeline: the reference's `I2VGenXLPipeline` (code/i2v_enhance/pipeline_i2vgen_xl.py) after `from_pretrained` / construction
    (i2v_enhance_interface.py:65-83).  Builds the MI355X mirror of its UNet from the same weights (strict: the 1511 keys of the vendored
    I2VGenXLUNet) and puts it in the UNet's place, so that the reference's unmodified `__call__` denoise loop (:841-913) calls
    `self.unet(latent_model_input, t, encoder_hidden_states=, fps=, image_latents=, image_embeddings=, cross_attention_kwargs=,
    return_dict=False, use_memopt=)[0]` (:857-867) on libsvdhip.so.  Returns the wrapped streamingt2v_amd.i2vgen_unet.I2VGenXLUNet.
    Executed by tests/test_dropin_enhancer_reference.py (CPU statements of the launchers, unmodified reference pipeline)."""
    from .i2vgen_unet import I2VConfig, I2VGenXLUNet
    old = pipeline.unet
    sd = state_dict if state_dict is not None else old.state_dict()
    if cfg is None:
        oc = getattr(old, "config", None)
        get = (lambda k, d: (oc[k] if isinstance(oc, dict) else getattr(oc, k)) if (oc is not None and (k in oc if isinstance(oc, dict) else hasattr(oc, k))) else d)
        boc = tuple(get("block_out_channels", (320, 640, 1280, 1280)))
        down = tuple(get("down_block_types", ("CrossAttnDownBlock3D",) * (len(boc) - 1) + ("DownBlock3D",)))
        cfg = I2VConfig(in_channels=get("in_channels", 4), out_channels=get("out_channels", 4), block_out_channels=boc,
                        layers_per_block=get("layers_per_block", 2), cross_attention_dim=get("cross_attention_dim", 1024),
                        attn_levels=tuple(t.startswith("CrossAttn") for t in down), sample_size=get("sample_size", None))
    unet = I2VGenXLUNet(cfg).load_state_dict(sd, device=device)
    shell = HipModule(unet)
    if hasattr(pipeline, "register_modules"):
        pipeline.register_modules(unet=shell)        # diffusers' DiffusionPipeline keeps its module registry in sync through this call
    else:
        pipeline.unet = shell
    return unet
