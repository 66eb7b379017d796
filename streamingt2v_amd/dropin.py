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
    shell = HipModule(wrapper)
    model.inference_model = shell
    model.first_stage_model.decoder = VideoDecoderModule(dec)       # AutoencodingEngine.decode() keeps calling self.decoder(z, timesteps=n)
    # The reference REBUILDS `self.inference_model = StreamingWrapper(self.model.diffusion_model, self.controlnet, ...)` at the start of every predict epoch
    # (`on_inference_epoch_start`, streaming_svd.py:46-55) -- which would silently put its own networks back.  The hook is wrapped so that the swap is
    # re-applied right after it, whatever the order of install() and the first trainer.predict().
    orig_hook = getattr(model, "on_inference_epoch_start", None)
    if callable(orig_hook) and not getattr(orig_hook, "_svdhip_reinstalls", False):
        def on_inference_epoch_start(*args, **kwargs):
            out = orig_hook(*args, **kwargs)
            model.inference_model = shell
            return out
        on_inference_epoch_start._svdhip_reinstalls = True
        object.__setattr__(model, "on_inference_epoch_start", on_inference_epoch_start)
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


class SvdPipelineMirror:
    """What the reference's `self.svd_pipeline` is called for (code/diffusion_trainer/streaming_svd.py:388-390):

        video_chunks = self.svd_pipeline(image, decode_chunk_size=decode_chunk_size).frames[0]          # list of 25 PIL frames

    i.e. diffusers' StableVideoDiffusionPipeline.__call__ with its defaults (576 x 1024, 25 frames, 25 steps, guidance 1 -> 3, fps 7, motion
    bucket 127, noise_aug_strength 0.02) -- on libsvdhip.so: `SVDConditioner.first_chunk` + `StreamingSVD._generate_initial_chunk` (the call's
    semantics, restated in diffusers' own formulation in oracle/svd_pipeline_oracle.py and compared in tests/test_host_svd_cpu.py).  The two
    random draws follow the pipeline's order: the image's noise augmentation first, then the initial latents (`generator`, else the global stream)."""

    def __init__(self, wrapper, first_stage_model, conditioner, num_frames=25, device="cuda"):
        from .sampling import EulerEDMSampler
        from .streaming_svd import StreamingSVD
        self.num_frames, self.device, self.conditioner = num_frames, device, conditioner
        self.svd = StreamingSVD(wrapper, first_stage_model, EulerEDMSampler(num_frames=num_frames))

    # What the reference calls on the pipeline object besides __call__: `post_init` (streaming_svd.py:58-62, fired by the first trainer.predict, i.e. AFTER an
    # install at the end of init_model) does `self.svd_pipeline.set_progress_bar_config(disable=True)` and, on a GPU, `enable_model_cpu_offload(gpu_id=...)`.
    # Neither has anything to act on here (no progress bar; the networks stay resident in 288 GB): accepted and ignored.
    def set_progress_bar_config(self, **kwargs):
        return None

    def enable_model_cpu_offload(self, gpu_id=None, device="cuda", **kwargs):
        return None

    def to(self, *args, **kwargs):
        return self

    def __call__(self, image, height=576, width=1024, num_frames=None, num_inference_steps=25, min_guidance_scale=1.0, max_guidance_scale=3.0,
                 fps=7, motion_bucket_id=127, noise_aug_strength=0.02, decode_chunk_size=None, generator=None, output_type="pil", **_ignored):
        import types
        import numpy as np
        import PIL.Image
        import torch
        T = num_frames or self.num_frames
        assert T == self.num_frames, "the mirror was built for the pipeline's own num_frames"
        if decode_chunk_size not in (4, 8):
            # diffusers decodes ALL frames in one group for None / num_frames; the temporal convolutions zero-pad at group boundaries, so the grouping is part of
            # the result -- only the two groupings the reference uses (8, or 4 with memopt: streaming_svd.py:127,388-390) are reproduced
            raise NotImplementedError(f"decode_chunk_size={decode_chunk_size!r}: the reference passes 8 (4 with memopt); other groupings change the decoded frames")
        if isinstance(image, PIL.Image.Image):
            if image.size != (width, height):
                image = image.resize((width, height), resample=PIL.Image.LANCZOS)          # VaeImageProcessor.preprocess(resample="lanczos")
            image = torch.from_numpy(np.asarray(image.convert("RGB")).copy()).permute(2, 0, 1).float() / 255.0
        img = (image.to(self.device, torch.float32) * 2.0 - 1.0).contiguous()                # [3, H, W] in [-1, 1]
        cond = self.conditioner
        cond.fps_id, cond.motion, cond.cond_aug = fps - 1, motion_bucket_id, noise_aug_strength   # "the model was trained on fps - 1"
        rdev = generator.device if generator is not None else self.device          # diffusers draws on the execution device when no generator is given
        aug = torch.randn((1,) + tuple(img.shape), generator=generator, device=rdev)
        noise = torch.randn((T, 4, height // 8, width // 8), generator=generator, device=rdev)
        c, uc = cond.first_chunk(img, aug_noise=aug)
        self.svd.use_memopt = decode_chunk_size == 4
        frames = self.svd._generate_initial_chunk(c, uc, noise.to(self.device), num_steps=num_inference_steps, min_scale=min_guidance_scale,
                                                  max_scale=max_guidance_scale)
        u8 = ((frames / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().cpu().numpy() * 255).round().astype("uint8")   # VaeImageProcessor.postprocess "pil"
        if output_type == "pil":
            return types.SimpleNamespace(frames=[[PIL.Image.fromarray(f) for f in u8]])
        return types.SimpleNamespace(frames=[u8])


def install_svd_pipeline(model, device="cuda"):
    """model: the reference's StreamingSVD module with its `svd_pipeline` (diffusers StableVideoDiffusionPipeline.from_pretrained(
    "stabilityai/stable-video-diffusion-img2vid-xt"), config.yaml:280-299) loaded.  Builds the MI355X networks from the PIPELINE'S OWN weights --
    `pipe.unet` / `pipe.vae` / `pipe.image_encoder` state dicts through the key maps of diffusers_keys.py (strict) -- and puts a
    `SvdPipelineMirror` in the pipeline's place, so that the reference's unmodified `image_to_video` (streaming_svd.py:359-402) generates chunk 0
    on libsvdhip.so with the stock SVD-XT weights.  `svd_pipeline` is a plain attribute (a DiffusionPipeline is not an nn.Module): assignable.

        from streamingt2v_amd import dropin
        dropin.install(self.model); dropin.install_svd_pipeline(self.model)         # end of inference_i2v.StreamingPipeline.init_model()
    """
    from .pipeline import stock_svd_xt_from_state_dicts
    pipe = model.svd_pipeline
    cfg = lambda m: dict(getattr(m, "config", {}) or {}) if not hasattr(getattr(m, "config", None), "to_dict") else m.config.to_dict()
    T = cfg(pipe.unet).get("num_frames", 25)
    wrapper, fsm, cond = stock_svd_xt_from_state_dicts(pipe.unet.state_dict(), cfg(pipe.unet), pipe.vae.state_dict(), cfg(pipe.vae),
                                                       pipe.image_encoder.state_dict(), cfg(pipe.image_encoder), device=device, num_frames=T,
                                                       num_conditional_frames=getattr(getattr(model, "inference_params", None), "num_conditional_frames", 7))
    model.svd_pipeline = SvdPipelineMirror(wrapper, fsm, cond, num_frames=T, device=device)
    return model.svd_pipeline
