"""`inference_i2v.py`-compatible front end: the call surface of the reference's ``StreamingPipeline``
(code/inference_i2v.py:52-258) on the MI355X path, without Lightning / jsonargparse.

    pipe = StreamingPipeline.from_checkpoint("model.safetensors", conditioner=...)       # init_model, :125-160
    video = pipe.image_to_video(image, num_frames=100)                                    # :179-190  -> uint8 [F, H, W, 3]
    video = pipe.enhance_video(image=image, video=video, use_randomized_blending=True)    # :192-207
    video = pipe.interpolate_video(video, dest_num_frames=200)                            # :209-223

What is native here: the checkpoint key map (same `state_dict` keys, strict), chunk arithmetic, the AR generation, the
enhancement loop with randomized blending and its key-frame pre-pass, range conversions.  What is an injected callable
(not on the measured path): the conditioners -- OpenCLIP image tower + VAE encoder for stage 1
(``conditioner(frame[3,H,W] in [-1,1]) -> (c, uc)``), CLIP text/vision + AutoencoderKL encode/decode for the enhancer
(``enhance_codec``), and ``vfi`` (`ema_vfi.EMAVFI`, the native EMA-VFI, or any callable).  Calling a stage whose callable was not supplied raises NotImplementedError that names it.
"""
import math
import random

import torch

# config.yaml values the reference's front end reads (inference_i2v.py:31-47, config.yaml:2,146-156, i2v_enhance_interface.py:82-131)
DEFAULTS = dict(num_frames=200, out_fps=24, chunk_size=38, overlap_size=12, use_randomized_blending=False, seed=33,
                num_frames_per_chunk=25, num_conditional_frames=7, num_steps=30, enhance_steps=30, enhance_strength=0.97,
                enhance_guidance_scale=9.0, enhance_target_fps=38, enhance_generator_seed=8888, enhance_height=720, enhance_width=1280,
                prompt="High Quality, HQ, detailed.",
                negative_prompt="Distorted, blurry, discontinuous, Ugly, blurry, low resolution, motionless, static, disfigured, "
                                "disconnected limbs, Ugly faces, incomplete arms")

CKPT_PREFIXES = dict(unet="model.diffusion_model.", controlnet="controlnet.", decoder="first_stage_model.decoder.",
                     clip="conditioner.embedders.0.open_clip.model.", cond_encoder="conditioner.embedders.3.encoder.")


def load_streamingsvd_checkpoint(path_or_state_dict, device="cuda", unet_cfg=None, vae_cfg=None):
    """PAIR/StreamingSVD ``model.safetensors`` (or a Lightning ``.ckpt``) -> (VideoUNet, ControlNet, VideoDecoder) on `device`.
    Strict: every key under the three prefixes must match our specs (= the reference modules' state_dicts); keys of the parts that
    stay on the reference side (``conditioner.*``, ``first_stage_model.encoder.*``, ``image_encoder_apm*``) are ignored here."""
    from .temporal_ae import VaeConfig, VideoDecoder
    from .video_model import ControlNet, UNetConfig, VideoUNet
    sd = path_or_state_dict
    if isinstance(sd, str):
        if sd.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(sd)
        else:
            sd = torch.load(sd, map_location="cpu")["state_dict"]
    cfg = unet_cfg or UNetConfig()
    unet = VideoUNet(cfg).load_state_dict(sd, device=device, prefix=CKPT_PREFIXES["unet"])
    cnet = ControlNet(cfg).load_state_dict(sd, device=device, prefix=CKPT_PREFIXES["controlnet"])
    dec = VideoDecoder(vae_cfg or VaeConfig()).load_state_dict(sd, device=device, prefix=CKPT_PREFIXES["decoder"])
    return unet, cnet, dec


def load_conditioner(state_dict, device="cuda", clip_cfg=None, vae_cfg=None, num_frames=25, generator=None):
    """The stage-1 conditioner from the same checkpoint: OpenCLIP ViT-H/14 image tower (``conditioner.embedders.0.open_clip.model.
    visual.*``) + cond-frame encoder (``conditioner.embedders.3.encoder.{encoder,quant_conv}.*``) -> conditioner.SVDConditioner.
    The text tower / logit_scale keys of open_clip and the embedder's decoder keys are ignored (never used by the reference)."""
    from .clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from .conditioner import SVDConditioner
    from .temporal_ae import CondFrameEncoder, VaeConfig
    clip = OpenCLIPVisionTower(clip_cfg or ClipVisionConfig()).load_state_dict(state_dict, device=device, prefix=CKPT_PREFIXES["clip"])
    enc = CondFrameEncoder(vae_cfg or VaeConfig()).load_state_dict(state_dict, device=device, prefix=CKPT_PREFIXES["cond_encoder"])
    return SVDConditioner(clip, enc, num_frames=num_frames, generator=generator)


def stock_svd_xt_from_state_dicts(usd, ucfg, vsd, vcfg, isd, icfg, device="cuda", num_frames=25, num_conditional_frames=7, generator=None):
    """The three networks of diffusers' StableVideoDiffusionPipeline from their diffusers-named state dicts + config dicts (the values of
    `pipe.unet.state_dict()` / `pipe.unet.config`, ... or of the folder's safetensors / config.json) -> (StreamingWrapper around the stock UNet,
    AutoencodingEngineDecoder around the stock decoder, SVDConditioner on the stock towers).  Key maps: diffusers_keys.py (strict both ways)."""
    from .clip_vision import ClipVisionConfig, OpenCLIPVisionTower, hf_clip_vision_to_openclip_keys
    from .conditioner import SVDConditioner
    from .diffusers_keys import svd_unet_diffusers_to_sgm, svd_vae_diffusers_to_sgm
    from .temporal_ae import AutoencodingEngineDecoder, CondFrameEncoder, VaeConfig, VideoDecoder
    from .video_model import UNetConfig, VideoUNet
    from .wrappers import StreamingWrapper
    g = lambda c, k, d: (c[k] if k in c else d) if isinstance(c, dict) else getattr(c, k, d)
    boc = tuple(g(ucfg, "block_out_channels", (320, 640, 1280, 1280)))
    down = g(ucfg, "down_block_types", ("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",))
    cfg = UNetConfig(in_channels=g(ucfg, "in_channels", 8), model_channels=boc[0], out_channels=g(ucfg, "out_channels", 4),
                     num_res_blocks=g(ucfg, "layers_per_block", 2), channel_mult=tuple(c // boc[0] for c in boc),
                     attention_resolutions=tuple(2 ** i for i, t in enumerate(down) if t.startswith("CrossAttn"))[::-1],
                     context_dim=g(ucfg, "cross_attention_dim", 1024), adm_in_channels=g(ucfg, "projection_class_embeddings_input_dim", 768),
                     controlnet_mode=False)
    unet = VideoUNet(cfg)
    unet.load_state_dict(svd_unet_diffusers_to_sgm(usd, unet.spec(), cfg.num_res_blocks), device=device)
    vb = tuple(g(vcfg, "block_out_channels", (128, 256, 512, 512)))
    vae_cfg = VaeConfig(vb[0], tuple(c // vb[0] for c in vb), g(vcfg, "layers_per_block", 2))
    dec, enc = VideoDecoder(vae_cfg), CondFrameEncoder(vae_cfg)
    dsd, esd = svd_vae_diffusers_to_sgm(vsd, dec.spec(), enc.spec(), len(vb))
    dec.load_state_dict(dsd, device=device)
    enc.load_state_dict(esd, device=device)
    iv = ClipVisionConfig(width=g(icfg, "hidden_size", 1280), layers=g(icfg, "num_hidden_layers", 32), heads=g(icfg, "num_attention_heads", 16),
                          patch_size=g(icfg, "patch_size", 14), image_size=g(icfg, "image_size", 224), embed_dim=g(icfg, "projection_dim", 1024),
                          mlp_ratio=g(icfg, "intermediate_size", 5120) / g(icfg, "hidden_size", 1280))
    isd = {k: v for k, v in isd.items() if "position_ids" not in k}
    tower = OpenCLIPVisionTower(iv).load_state_dict(hf_clip_vision_to_openclip_keys(isd, iv.layers), device=device)
    nf = g(ucfg, "num_frames", num_frames)
    return (StreamingWrapper(unet, None, num_conditional_frames), AutoencodingEngineDecoder(dec),
            SVDConditioner(tower, enc, num_frames=nf, generator=generator))


def load_stock_svd_xt(folder, device="cuda", variant="fp16", num_frames=25, num_conditional_frames=7, generator=None):
    """The networks of CHUNK 0 from a diffusers-format ``stabilityai/stable-video-diffusion-img2vid-xt`` folder -- what the reference's
    ``svd_pipeline`` module is (config.yaml:280-299: StableVideoDiffusionPipeline.from_pretrained(..., torch_dtype=float16, variant="fp16"), called
    at streaming_svd.py:388-390): unet/ (UNetSpatioTemporalConditionModel = the sgm VideoUNet without ControlNet / CAM, re-keyed), vae/
    (AutoencoderKLTemporalDecoder = sgm Encoder + VideoDecoder, re-keyed), image_encoder/ (CLIPVisionModelWithProjection).  Returns
    (StreamingWrapper around the stock UNet, AutoencodingEngineDecoder around the stock decoder, SVDConditioner on the stock towers) for
    ``StreamingSVD.set_initial_model``.  Key maps: diffusers_keys.py (names restated from diffusers 0.30.2, strict in both directions).
    Inside the reference itself the same networks come from the loaded pipeline object: dropin.install_svd_pipeline."""
    import json
    import os
    from safetensors.torch import load_file

    def part(name, stem):
        d = os.path.join(folder, name)
        cfg = json.load(open(os.path.join(d, "config.json")))
        for fn in (f"{stem}.{variant}.safetensors", f"{stem}.safetensors"):
            if os.path.exists(os.path.join(d, fn)):
                return cfg, load_file(os.path.join(d, fn))
        raise FileNotFoundError(f"no {stem}[.{variant}].safetensors under {d}")

    ucfg, usd = part("unet", "diffusion_pytorch_model")
    vcfg, vsd = part("vae", "diffusion_pytorch_model")
    icfg, isd = part("image_encoder", "model")
    return stock_svd_xt_from_state_dicts(usd, ucfg, vsd, vcfg, isd, icfg, device=device, num_frames=num_frames,
                                         num_conditional_frames=num_conditional_frames, generator=generator)


def load_enhancer(folder, device="cuda", variant="fp16", generator=None):
    """The enhancement stage from a diffusers-format ``ali-vilab/i2vgen-xl`` folder (i2v_enhance_interface.py:63-80 loads it with
    I2VGenXLPipeline.from_pretrained(..., torch_dtype=float16, variant="fp16")): unet/, vae/, text_encoder/, image_encoder/ with their
    config.json -> (I2VGenXLUNet, EnhanceCodec).  Weights are read from ``*.safetensors`` (``.<variant>`` tried first)."""
    import json
    import os
    from safetensors.torch import load_file
    from .clip_text import CLIPTextTower, ClipTextConfig
    from .clip_vision import ClipVisionConfig, OpenCLIPVisionTower, hf_clip_vision_to_openclip_keys
    from .enhance_codec import EnhanceCodec
    from .i2vgen_unet import I2VConfig, I2VGenXLUNet
    from .temporal_ae import AutoencoderKL2D, VaeConfig

    def part(name, stem):
        d = os.path.join(folder, name)
        cfg = json.load(open(os.path.join(d, "config.json")))
        for fn in (f"{stem}.{variant}.safetensors", f"{stem}.safetensors"):
            if os.path.exists(os.path.join(d, fn)):
                return cfg, load_file(os.path.join(d, fn))
        raise FileNotFoundError(f"no {stem}[.{variant}].safetensors under {d}")

    ucfg, usd = part("unet", "diffusion_pytorch_model")
    attn = tuple(t.startswith("CrossAttn") for t in ucfg.get("down_block_types", ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",)))
    unet = I2VGenXLUNet(I2VConfig(block_out_channels=tuple(ucfg.get("block_out_channels", (320, 640, 1280, 1280))),
                                  layers_per_block=ucfg.get("layers_per_block", 2), cross_attention_dim=ucfg.get("cross_attention_dim", 1024),
                                  attn_levels=attn)).load_state_dict(usd, device=device)
    vcfg, vsd = part("vae", "diffusion_pytorch_model")
    boc = vcfg.get("block_out_channels", (128, 256, 512, 512))
    vae = AutoencoderKL2D(VaeConfig(boc[0], tuple(c // boc[0] for c in boc), vcfg.get("layers_per_block", 2)),
                          scaling_factor=vcfg.get("scaling_factor", 0.18215)).load_state_dict(vsd, device=device, diffusers_keys=True)
    icfg, isd = part("image_encoder", "model")
    iv = ClipVisionConfig(width=icfg["hidden_size"], layers=icfg["num_hidden_layers"], heads=icfg["num_attention_heads"],
                          patch_size=icfg["patch_size"], image_size=icfg["image_size"], embed_dim=icfg["projection_dim"],
                          mlp_ratio=icfg["intermediate_size"] / icfg["hidden_size"])
    tower = OpenCLIPVisionTower(iv).load_state_dict(hf_clip_vision_to_openclip_keys(isd, iv.layers), device=device)
    tcfg, tsd = part("text_encoder", "model")
    tc = ClipTextConfig(vocab_size=tcfg["vocab_size"], hidden_size=tcfg["hidden_size"], intermediate_size=tcfg["intermediate_size"],
                        num_hidden_layers=tcfg["num_hidden_layers"], num_attention_heads=tcfg["num_attention_heads"],
                        max_position_embeddings=tcfg["max_position_embeddings"])
    tsd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in tsd.items()}
    text = CLIPTextTower(tc).load_state_dict(tsd, device=device)
    gen = generator
    if gen is None and str(device).startswith("cuda"):
        gen = torch.Generator(device=device).manual_seed(DEFAULTS["enhance_generator_seed"])       # torch.manual_seed(8888), interface :64
    codec = EnhanceCodec(vae, tower, text, generator=gen, device=device)
    sched_cfg = os.path.join(folder, "scheduler", "scheduler_config.json")
    if os.path.exists(sched_cfg):                                   # the checkpoint's own DDIM configuration, not the recalled defaults
        from .enhance import DDIMSchedule
        with open(sched_cfg) as f:
            codec.scheduler = DDIMSchedule.from_config(json.load(f))
    tok_dir = os.path.join(folder, "tokenizer")
    if os.path.exists(os.path.join(tok_dir, "vocab.json")):        # the pipeline's CLIPTokenizer files: prompts of i2v_enhance_interface.py:99-100
        from .clip_tokenizer import CLIPBPETokenizer
        codec.set_prompts(DEFAULTS["prompt"], DEFAULTS["negative_prompt"], CLIPBPETokenizer.from_pretrained(tok_dir))
    return unet, codec


def load_vfi(path_or_state_dict, device="cuda", cfg=None):
    """EMA-VFI from the reference's ``ours.pkl`` (i2v_enhance_interface.vfi_init :15-27: init_model_config(F=32, depth=[2,2,2,4,4]),
    Trainer.Model.load_model strips ``module.`` and drops the cached attn_mask / HW buffers)."""
    from .ema_vfi import EMAVFI
    sd = path_or_state_dict
    if isinstance(sd, str):
        sd = torch.load(sd, map_location="cpu")
    if any(k.startswith("module.") for k in sd):
        sd = EMAVFI.convert_checkpoint(sd)
    return EMAVFI(cfg).load_state_dict(sd, device=device)


def resize_and_keep(image, height=576):
    """utils/inference_utils.py:36-41, applied to the input image by the trainer's image_to_video (streaming_svd.py:381-383): PIL default
    (BICUBIC) resize to `height` rows keeping the aspect ratio, width truncated to an int.  uint8 [H, W, 3] / PIL in, uint8 array out."""
    import numpy as np
    import PIL.Image
    img = image if isinstance(image, PIL.Image.Image) else PIL.Image.fromarray(np.asarray(image))
    wsize = int(float(img.size[0]) * (height / float(img.size[1])))
    return np.asarray(img.resize((wsize, height)))


def resize_key_image(image, width=1280, height=720):
    """inference_i2v.py:193-194: the key image enters the enhancer as IImage(image).resize((720, 1280)) -- PIL BICUBIC to width x height
    (skipped when it already has that size, lib/farancia/libimage/iimage.py:169-192) -- before `_center_crop_wide` sees it.  Already
    enhanced key frames (the second stage of randomized blending) have the target size and pass through."""
    import numpy as np
    import PIL.Image
    img = image if isinstance(image, PIL.Image.Image) else PIL.Image.fromarray(np.asarray(image))
    return img if img.size == (width, height) else img.resize((width, height), resample=PIL.Image.BICUBIC)


def num_autoregressive_generations(num_frames, frames_per_chunk=25, num_conditional_frames=7):
    """inference_i2v.py:182-186."""
    return max(0, math.ceil((num_frames - frames_per_chunk) / (frames_per_chunk - num_conditional_frames)))


def enhance_windows(n_frames, chunk_size, overlap_size):
    """Window starts of the randomized-blending pass and the number of frames that survive (i2v_enhance_interface.py:88-113):
    full windows only, stride chunk - overlap; the tail that does not fill a window is dropped."""
    starts = [i for i in range(0, n_frames, chunk_size - overlap_size) if i + chunk_size <= n_frames]
    max_idx = (chunk_size - overlap_size) * (len(starts) - 1) + chunk_size if starts else 0
    return starts, max_idx


class StreamingPipeline:
    def __init__(self, unet, controlnet, decoder, conditioner=None, enhancer_unet=None, enhance_codec=None, vfi=None, **overrides):
        from .sampling import AlignYourSteps, EulerEDMSampler
        from .streaming_svd import StreamingSVD
        from .temporal_ae import AutoencodingEngineDecoder
        from .wrappers import StreamingWrapper
        self.cfg = dict(DEFAULTS, **overrides)
        c = self.cfg
        sampler = EulerEDMSampler(num_steps=c["num_steps"], num_frames=c["num_frames_per_chunk"], min_scale=1.5, max_scale=3.0,
                                  discretization=AlignYourSteps())
        self.model = StreamingSVD(StreamingWrapper(unet, controlnet, c["num_conditional_frames"]), AutoencodingEngineDecoder(decoder),
                                  sampler, num_conditional_frames=c["num_conditional_frames"])
        self.conditioner, self.enhancer_unet, self.enhance_codec, self.vfi = conditioner, enhancer_unet, enhance_codec, vfi
        self.num_frames, self.out_fps = c["num_frames"], c["out_fps"]
        self.use_randomized_blending, self.chunk_size, self.overlap_size = c["use_randomized_blending"], c["chunk_size"], c["overlap_size"]
        self.device = unet.device

    @classmethod
    def from_pretrained(cls, streamingsvd_ckpt, i2vgen_folder=None, vfi_ckpt=None, device="cuda", svd_xt_folder=None, **kw):
        """Everything inference_i2v.StreamingPipeline.init_model assembles (:125-165), from the same artefacts: the StreamingSVD
        checkpoint (UNet + ControlNet + decoder + stage-1 conditioner), the diffusers-format i2vgen-xl folder, EMA-VFI's ours.pkl, and
        -- svd_xt_folder -- the stock stabilityai/stable-video-diffusion-img2vid-xt folder the reference generates CHUNK 0 with
        (config.yaml:280-299; load_stock_svd_xt).  Without it chunk 0 runs on the StreamingSVD checkpoint's own UNet / decoder."""
        sd = streamingsvd_ckpt
        if isinstance(sd, str):
            if sd.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd = load_file(sd)
            else:
                sd = torch.load(sd, map_location="cpu")["state_dict"]
        cfgs = {k: kw.pop(k, None) for k in ("unet_cfg", "vae_cfg", "clip_cfg", "cond_vae_cfg", "vfi_cfg")}   # non-default sizes (tests)
        kw.setdefault("conditioner", load_conditioner(sd, device=device, clip_cfg=cfgs["clip_cfg"], vae_cfg=cfgs["cond_vae_cfg"],
                                                      num_frames=kw.get("num_frames_per_chunk", DEFAULTS["num_frames_per_chunk"])))
        if i2vgen_folder is not None:
            kw["enhancer_unet"], kw["enhance_codec"] = load_enhancer(i2vgen_folder, device=device)
        if vfi_ckpt is not None:
            kw["vfi"] = load_vfi(vfi_ckpt, device=device, cfg=cfgs["vfi_cfg"])
        kw.setdefault("input_height", 576)
        pipe = cls(*load_streamingsvd_checkpoint(sd, device=device, unet_cfg=cfgs["unet_cfg"], vae_cfg=cfgs["vae_cfg"]), **kw)
        if svd_xt_folder is not None:
            pipe.model.set_initial_model(*load_stock_svd_xt(svd_xt_folder, device=device, num_frames=pipe.cfg["num_frames_per_chunk"],
                                                            num_conditional_frames=pipe.cfg["num_conditional_frames"]))
        return pipe

    @classmethod
    def from_checkpoint(cls, path, device="cuda", **kw):
        kw.setdefault("input_height", 576)
        return cls(*load_streamingsvd_checkpoint(path, device=device), **kw)

    # ------------------------------------------------------------------------------------------ stage 1
    def image_to_video(self, image, num_frames, seed=None, **kwargs):
        """image: uint8 [H, W, 3] (numpy or tensor; with `input_height=576` -- what `from_checkpoint` configures -- it is brought to 576 rows
        by `resize_and_keep` and must then be 1024 wide, like streaming_svd.py:381-384) or fp32 [3, H, W] in [-1, 1] (used as is).
        Returns uint8 [num_frames, H, W, 3] like trainer.generated_video[:num_frames]."""
        if self.conditioner is None:
            raise NotImplementedError("image_to_video needs conditioner(frame) -> (c, uc): the OpenCLIP image tower + VAE encoder of "
                                      "the reference's GeneralConditioner are not part of the MI355X path (SURVEY.md 8f N4)")
        img = torch.as_tensor(image)
        if img.dtype == torch.uint8:
            if self.cfg.get("input_height"):                                     # 576 for the shipped model (from_checkpoint's default)
                img = torch.as_tensor(resize_and_keep(img.cpu().numpy(), self.cfg["input_height"]).copy())
                assert img.shape[1] == self.cfg["input_height"] * 16 // 9, f"input image must be 16:9 (got {tuple(img.shape)} after resize_and_keep)"
            img = img.to(self.device).permute(2, 0, 1).float() / 127.5 - 1.0
        img = img.to(self.device, torch.float32)
        c = self.cfg
        T, h, w = c["num_frames_per_chunk"], img.shape[1] // 8, img.shape[2] // 8
        n_ar = num_autoregressive_generations(num_frames, T, c["num_conditional_frames"])
        g = torch.Generator(device=self.device)
        g.manual_seed(c["seed"] if seed is None else seed)                       # seed_everything: 33 (config.yaml:2)
        noises = [torch.randn(T, 4, h, w, generator=g, device=self.device) for _ in range(1 + n_ar)]
        video = self.model.image_to_video(self.conditioner, img, num_frames, noises)
        return self.model.to_uint8_video(video).cpu().numpy()

    # ------------------------------------------------------------------------------------------ enhancement
    def enhance_video(self, image, video, chunk_size=38, overlap_size=12, strength=0.97, use_randomized_blending=False, rng=None, seed=None,
                      **kwargs):
        """Mirror of enhance_video + i2v_enhance_process (inference_i2v.py:192-207, i2v_enhance_interface.py:82-133).
        Blending offsets: the reference draws them from Python's GLOBAL `random` stream, seeded once by Lightning's seed_everything
        (config.yaml:2; pipeline_i2vgen_xl.py:896).  Here the stream is explicit: `rng` (a random.Random, e.g. one shared by several calls
        to reproduce the reference's single global stream) or `seed` (default: the configured seed, i.e. every call draws the same offsets).
        enhance_codec: enhance_codec.EnhanceCodec (native: AutoencoderKL2D + CLIP towers) or any object with encode_video(frames) ->
        latents [1,4,F,90,160], window_conditioning(images, n_windows, window_len) -> list of dicts(fps, image_latents,
        image_embeddings, text) with the unconditional half first, noise_like(latents), decode(latents) -> uint8 [F,H,W,3]."""
        if self.enhancer_unet is None or self.enhance_codec is None:
            raise NotImplementedError("enhance_video needs enhancer_unet (I2VGenXLUNet) and enhance_codec (CLIP text/vision towers + "
                                      "AutoencoderKL encode/decode of the I2VGen-XL pipeline; SURVEY.md 8f N4)")
        from .enhance import I2VEnhancer
        c, codec = self.cfg, self.enhance_codec
        enh = I2VEnhancer(self.enhancer_unet, getattr(codec, "scheduler", None), guidance_scale=c["enhance_guidance_scale"],
                          num_inference_steps=c["enhance_steps"], strength=strength)
        if kwargs:
            raise TypeError(f"enhance_video got unexpected keyword arguments {sorted(kwargs)}")
        if rng is None:
            rng = random.Random(c["seed"] if seed is None else seed)
        video = list(video)
        dist_kw = dict(group=getattr(self, "group", None))
        if getattr(self, "plan", None) is not None:
            dist_kw["plan"] = self.plan
        images = [resize_key_image(image, getattr(codec, "w", c["enhance_width"]), getattr(codec, "h", c["enhance_height"]))]
        if use_randomized_blending:
            starts, max_idx = enhance_windows(len(video), chunk_size, overlap_size)
            key_frames = [video[s] for s in starts]                              # 1st frame of every window, enhanced first
            conds = codec.window_conditioning(images, 1, len(key_frames))         # the reference's order of random draws: image latents,
            lat = codec.encode_video(key_frames)                                 # video posterior, SDEdit noise (pipeline_i2vgen_xl.py:784-829)
            images = list(codec.decode(enh.denoise(lat, codec.noise_like(lat), conds, len(key_frames), 0, rng, **dist_kw)))   # one window: its two CFG halves on two ranks
            video = video[:max_idx]
        else:
            starts, chunk_size, overlap_size = [0], len(video), 0
        conds = codec.window_conditioning(images, len(starts), chunk_size)
        lat = codec.encode_video(video)
        # self.group (a torch.distributed group, default None): the (window, CFG half) units of every DDIM step are sharded over its ranks
        # (blending.blend_step_units_sharded: identical offsets on every rank, one all-gather of the predictions per step);
        # self.plan (a parallel.JobPlan in "job" mode, default None; takes precedence): CFG pair x frame <-> pixel sequence parallelism inside every
        # window's UNet evaluation -- all ranks busy whatever the number of windows (I2VEnhancer.denoise)
        return codec.decode(enh.denoise(lat, codec.noise_like(lat), conds, chunk_size, overlap_size, rng, **dist_kw))

    def interpolate_video(self, video, dest_num_frames, **kwargs):
        """inference_i2v.py:211-224.  vfi: an `ema_vfi.EMAVFI` (the native EMA-VFI) or any callable vfi(video, dest_num_frames)."""
        if self.vfi is None:
            raise NotImplementedError("interpolate_video needs vfi: ema_vfi.EMAVFI().load_state_dict(EMAVFI.convert_checkpoint(torch.load('ours.pkl'))) "
                                      "(code/i2v_enhance/thirdparty/VFI, SURVEY.md 8f N4) or a callable vfi(video, dest_num_frames)")
        from .ema_vfi import EMAVFI, vfi_process
        if isinstance(self.vfi, EMAVFI):
            import numpy as np
            grp = getattr(self, "group", None)
            frames = vfi_process(list(video), self.vfi, dest_num_frames, device=self.vfi.device, group=grp, sharded=grp is not None)
            return np.stack([np.asarray(f) for f in frames], axis=0)
        return self.vfi(video, dest_num_frames)

    def __call__(self, image):
        """The script body of inference_i2v.py:226-258 for one image."""
        n, rb = self.num_frames, self.use_randomized_blending
        chunk, overlap = (self.chunk_size, self.overlap_size) if rb else ((n + 1) // 2, 0)
        video = self.image_to_video(image, (n + 1) // 2)
        video = self.enhance_video(image=image, video=video, use_randomized_blending=rb, chunk_size=chunk, overlap_size=overlap)
        return self.interpolate_video(video, dest_num_frames=n)
