"""TEST INFRASTRUCTURE (CPU oracle; never imported by the product path).

Chunk 0 as the REFERENCE computes it: code/diffusion_trainer/streaming_svd.py:388-390 calls
``self.svd_pipeline(image, decode_chunk_size=8).frames[0]`` with svd_pipeline = diffusers ``StableVideoDiffusionPipeline`` on the stock
``stabilityai/stable-video-diffusion-img2vid-xt`` fp16 weights (code/config.yaml:280-299).  diffusers==0.30.2 (requirements.txt:6) is an
un-vendored third-party dependency that is not installed here and cannot be fetched: this file RESTATES the call-level semantics of

    pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py   StableVideoDiffusionPipeline.__call__, _encode_image,
                                                                          _encode_vae_image, _get_add_time_ids, decode_latents,
                                                                          _resize_with_antialiasing, _gaussian_blur2d
    schedulers/scheduling_euler_discrete.py                               EulerDiscreteScheduler (SVD-XT's scheduler_config.json:
                                                                          v_prediction, use_karras_sigmas, sigma 0.002 .. 700,
                                                                          timestep_type "continuous", timestep_spacing "leading")
    image_processor.py                                                    VaeImageProcessor.postprocess("pil") -> uint8

IN DIFFUSERS' OWN FORMULATION (scale_model_input / pred_original_sample / derivative, numpy float64 Karras ramp, per-frame guidance
linspace(1, 3), 0.02 * randn noise augmentation, un-scaled posterior-mode image latents, added_time_ids (fps - 1, 127, 0.02)) -- not in the
sgm EDM formulation the product's sampler uses -- so that tests/test_host_svd_cpu.py::test_initial_chunk_follows_the_diffusers_pipeline checks
one against the other.  **Parity unpinned**: there is no diffusers here to pin this restatement to, and the reference holds no fixture of the call.

The networks are abstract callables with diffusers' signatures:
    image_encoder(pixel_values [1, 3, 224, 224])            -> image_embeds [1, 1024]        (CLIPVisionModelWithProjection)
    vae_encode_mode(image [1, 3, H, W])                      -> posterior mode [1, 4, h, w]   (vae.encode(x).latent_dist.mode(), NOT scaled)
    unet(sample [B, T, 8, h, w], t, encoder_hidden_states [B, 1, 1024], added_time_ids [B, 3]) -> [B, T, 4, h, w]
    vae_decode(z [n, 4, h, w], num_frames=n)                 -> [n, 3, H, W]
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ---------------------------------------------------------------------------------------------------------------- image pre-processing
def _gaussian(window_size, sigma):
    x = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    if window_size % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def _filter2d(x, kernel):
    """pipeline_stable_video_diffusion._filter2d: reflect padding, depth-wise correlation with one [kh, kw] kernel."""
    c = x.shape[1]
    kh, kw = kernel.shape
    xp = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2), mode="reflect")
    return F.conv2d(xp, kernel[None, None].expand(c, 1, kh, kw), groups=c)


def resize_with_antialiasing(x, size, interpolation="bicubic", align_corners=True):
    """_resize_with_antialiasing: Gaussian blur with sigma = max((factor - 1) / 2, 0.001) per axis, kernel int(max(4 sigma, 3)) made odd,
    separable (x pass then y pass) -- ALWAYS applied, unlike kornia's resize, which blurs only when down-scaling -- then F.interpolate."""
    h, w = x.shape[-2:]
    factors = (h / size[0], w / size[1])
    sigmas = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * sigmas[0], 3)), int(max(2.0 * 2 * sigmas[1], 3))]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    out_x = _filter2d(x, _gaussian(ks[1], sigmas[1])[None, :])
    out = _filter2d(out_x, _gaussian(ks[0], sigmas[0])[:, None])
    return F.interpolate(out, size=tuple(size), mode=interpolation, align_corners=align_corners)


def encode_image(image01, image_encoder):
    """_encode_image for a PIL input: [0, 1] -> [-1, 1] -> antialiased 224 x 224 -> [0, 1] -> CLIP mean / std (the feature extractor with
    do_resize = do_center_crop = do_rescale = False) -> image_embeds[:, None]; CFG: (zeros | embeds)."""
    x = image01 * 2.0 - 1.0
    x = resize_with_antialiasing(x, (224, 224))
    x = (x + 1.0) / 2.0
    mean, std = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    emb = image_encoder((x - mean) / std)[:, None]
    return torch.cat([torch.zeros_like(emb), emb])


# ---------------------------------------------------------------------------------------------------------------- scheduler
class EulerDiscreteKarras:
    """EulerDiscreteScheduler as SVD-XT configures it.  set_timesteps: Karras ramp in numpy float64 between sigma_max / sigma_min of the
    config, cast to fp32, 0 appended; timesteps = 0.25 ln(sigma); init_noise_sigma = sqrt(sigma_max^2 + 1) ("leading" spacing)."""

    def __init__(self, sigma_min=0.002, sigma_max=700.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def set_timesteps(self, n):
        ramp = np.linspace(0, 1, n)
        lo, hi = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        sig = (hi + ramp * (lo - hi)) ** self.rho
        self.timesteps = torch.from_numpy(np.array([0.25 * np.log(s) for s in sig])).to(torch.float32)
        self.sigmas = torch.cat([torch.from_numpy(sig).to(torch.float32), torch.zeros(1)])
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)
        self.i = 0

    def scale_model_input(self, sample):
        return sample / ((self.sigmas[self.i] ** 2 + 1) ** 0.5)

    def step(self, model_output, sample):
        """gamma = 0 (s_churn 0): sigma_hat = sigma; v_prediction: x0 = v * (-sigma / sqrt(sigma^2 + 1)) + sample / (sigma^2 + 1);
        derivative = (sample - x0) / sigma; prev = sample + derivative * (sigma_next - sigma).  fp32 like the scheduler's upcast."""
        s = self.sigmas[self.i]
        sample = sample.to(torch.float32)
        x0 = model_output * (-s / (s ** 2 + 1) ** 0.5) + sample / (s ** 2 + 1)
        d = (sample - x0) / s
        self.i += 1
        return sample + d * (self.sigmas[self.i] - s)


# ---------------------------------------------------------------------------------------------------------------- the call
def svd_pipeline_call(image01, image_encoder, vae_encode_mode, unet, vae_decode, *, aug_noise, latents, num_frames=25, num_inference_steps=25,
                      min_guidance_scale=1.0, max_guidance_scale=3.0, fps=7, motion_bucket_id=127, noise_aug_strength=0.02, decode_chunk_size=8,
                      scaling_factor=0.18215):
    """StableVideoDiffusionPipeline.__call__(image, decode_chunk_size=8) with its defaults.  image01 [1, 3, H, W] in [0, 1] (the PIL image);
    aug_noise [1, 3, H, W] and latents [1, T, 4, h, w] stand for the two randn_tensor draws (N(0, 1)).  Returns (uint8 frames [T, H, W, 3] =
    the PIL frames of `.frames[0]`, the final latents)."""
    emb = encode_image(image01, image_encoder)                                           # 3. [2, 1, 1024]
    fps = fps - 1                                                                        #    "the model was trained on fps - 1"
    image = 2.0 * image01 - 1.0                                                          # 4. VaeImageProcessor.preprocess (already 576 x 1024)
    image = image + noise_aug_strength * aug_noise
    lat = vae_encode_mode(image)
    image_latents = torch.cat([torch.zeros_like(lat), lat])[:, None].repeat(1, num_frames, 1, 1, 1)
    added_time_ids = torch.tensor([[fps, motion_bucket_id, noise_aug_strength]], dtype=torch.float32).repeat(2, 1)       # 5.
    sch = EulerDiscreteKarras()                                                          # 6.
    sch.set_timesteps(num_inference_steps)
    latents = latents * sch.init_noise_sigma                                             # 7. prepare_latents
    gs = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames)[None, :, None, None, None]                 # 8.
    for t in sch.timesteps:                                                              # 9.
        inp = sch.scale_model_input(torch.cat([latents] * 2))
        inp = torch.cat([inp, image_latents], dim=2)
        pred = unet(inp, t, emb, added_time_ids)
        pu, pc = pred.chunk(2)
        latents = sch.step(pu + gs * (pc - pu), latents)
    z = latents.flatten(0, 1) / scaling_factor                                           # decode_latents: groups of decode_chunk_size frames
    frames = torch.cat([vae_decode(z[i:i + decode_chunk_size], num_frames=z[i:i + decode_chunk_size].shape[0]) for i in range(0, z.shape[0], decode_chunk_size)])
    u8 = ((frames / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().numpy() * 255).round().astype("uint8")           # postprocess_video -> PIL
    return torch.from_numpy(u8), latents


def frames_back_to_float(u8):
    """code/diffusion_trainer/streaming_svd.py:392-393: torch.stack([ToTensor()(frame) ...]) * 2.0 - 1."""
    return u8.permute(0, 3, 1, 2).float() / 255.0 * 2.0 - 1


# ---------------------------------------------------------------------------------------------------------------- adaptors onto oracle/svd_oracle.py
def sgm_unet_as_diffusers(video_unet_fn, T):
    """UNetSpatioTemporalConditionModel.forward on top of a function with the sgm VideoUNet signature f(x [B*T, 8, h, w], timesteps [B*T],
    context [B*T, 1, 1024], y [B*T, 768]): the timestep is broadcast, the added time ids go through 256-wide sinusoids (add_time_proj: cos
    first, flip_sin_to_cos) and are concatenated, both embeddings and the context are repeated per frame (repeat_interleave)."""
    def pe(v, dim=256):
        half = dim // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        a = v.float()[:, None] * freqs[None]
        return torch.cat([torch.cos(a), torch.sin(a)], -1)

    def unet(sample, t, encoder_hidden_states, added_time_ids):
        B = sample.shape[0]
        y = pe(added_time_ids.flatten()).reshape(B, -1)
        out = video_unet_fn(sample.flatten(0, 1), t.expand(B).repeat_interleave(T), encoder_hidden_states.repeat_interleave(T, 0), y.repeat_interleave(T, 0))
        return out.reshape(B, T, *out.shape[1:])
    return unet
