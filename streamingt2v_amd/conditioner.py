"""Stage-1 conditioner on the MI355X kernels: what `GeneralConditioner` produces for one conditioning frame in
StreamingSVD._generate_conditional_output (code/diffusion_trainer/streaming_svd.py:166-199, embedders of config.yaml:160-218):

    crossattn [T, 1, 1024]  OpenCLIP ViT-H/14 image embedding of the frame            (clip_vision.OpenCLIPVisionTower)
    concat    [T, 4, h, w]  AutoencoderKLModeOnly.encode(frame + cond_aug * U[0,1) noise), scale 1   (temporal_ae.CondFrameEncoder)
    vector    [T, 768]      sinusoidal embeddings (256 each) of fps_id 6, motion_bucket_id 127, cond_aug 0.02, in that order
    uc        crossattn and concat zeroed (force_uc_zero_embeddings), vector unchanged

The 224 x 224 CLIP resize is kornia's ``geometry.resize(..., "bicubic", align_corners=True, antialias=True)`` (modules.py:624-636).
kornia==0.7.2 is neither vendored nor installed: `kornia_resize_antialias` restates its published algorithm (Gaussian blur with
sigma = (factor - 1) / 2 per axis, kernel size 4 sigma made odd, reflect border, separable; then F.interpolate bicubic) -- host-side image
preprocessing of ONE frame per chunk, **parity unpinned** against kornia itself.
"""
import math

import torch
import torch.nn.functional as F

from .clip_vision import CLIP_MEAN, CLIP_STD


def sinusoid(values, dim=256, max_period=10000.0):
    """sgm timestep_embedding (util.py:207-231): [cos | sin] of value * max_period^(-i/half); values [n] -> [n, dim] fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = torch.as_tensor(values, dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], -1)


def _gaussian_kernel1d(ks, sigma, device):
    x = torch.arange(ks, device=device, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def kornia_resize_antialias(x, size, interpolation="bicubic", align_corners=True, always_blur=False):
    """kornia.geometry.transform.resize(x, size, interpolation, align_corners, antialias=True) of kornia 0.7.2 (affwarp.py): blur only when
    downscaling; sigmas = max((factor - 1) / 2, 0.001); kernel = int(max(4 sigma, 3)) made odd; gaussian_blur2d(border 'reflect', separable).
    always_blur=True is diffusers' copy of it, `_resize_with_antialiasing` (pipeline_stable_video_diffusion.py, the CLIP resize of CHUNK 0,
    streaming_svd.py:390): the same blur applied unconditionally, no same-size shortcut.  576 x 1024 -> 224 x 224 takes the same path in both."""
    h, w = x.shape[-2:]
    if (h, w) == tuple(size) and not always_blur:
        return x
    factors = (h / size[0], w / size[1])
    if max(factors) > 1 or always_blur:
        sig = (max((factors[0] - 1.0) / 2.0, 0.001), max((factors[1] - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        C = x.shape[1]
        ky, kx = _gaussian_kernel1d(ks[0], sig[0], x.device), _gaussian_kernel1d(ks[1], sig[1], x.device)
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode="reflect")
        xp = F.conv2d(xp, kx.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)             # filter2d_separable: x pass, then y pass
        x = F.conv2d(xp, ky.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return F.interpolate(x, size=tuple(size), mode=interpolation, align_corners=align_corners)


class SVDConditioner:
    def __init__(self, clip_tower, cond_encoder, num_frames=25, fps_id=6, motion_bucket_id=127, cond_aug=0.02, generator=None):
        self.clip, self.enc, self.T = clip_tower, cond_encoder, num_frames
        self.fps_id, self.motion, self.cond_aug, self.gen = fps_id, motion_bucket_id, cond_aug, generator

    @staticmethod
    def clip_preprocess(img, always_blur=False):
        """[n, 3, H, W] in [-1, 1] -> CLIP-normalised [n, 3, 224, 224] (modules.py:624-636; always_blur: diffusers' _encode_image)."""
        x = kornia_resize_antialias(img.float(), (224, 224), "bicubic", align_corners=True, always_blur=always_blur)
        x = (x + 1.0) / 2.0
        mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    @torch.no_grad()
    def first_chunk(self, frame, aug_noise=None):
        """Conditioning of CHUNK 0 the way the reference computes it -- through diffusers' StableVideoDiffusionPipeline.__call__
        (streaming_svd.py:388-390; restated in oracle/svd_pipeline_oracle.py): CLIP embedding of the clean image resized with
        `_resize_with_antialiasing`; image latents = posterior MODE of the image + 0.02 * N(0, 1) noise (NORMAL, where the AR chunks'
        sgm conditioner adds 0.02 * U[0, 1)), not scaled; added_time_ids (fps - 1 = 6, motion_bucket_id 127, noise_aug_strength 0.02) -- the same
        three sinusoids as the AR chunks' `vector`; negative branch = zero embedding and zero latents."""
        return self(frame, _diffusers=True, aug_noise=aug_noise)

    @torch.no_grad()
    def __call__(self, frame, _diffusers=False, aug_noise=None):
        """frame [3, H, W] fp32 in [-1, 1] (on the GPU) -> (c, uc) dicts as StreamingSVD._generate_conditional_output consumes them."""
        T = self.T
        img = frame[None].float()
        dev = img.device
        pre = self.clip_preprocess(img, always_blur=True) if _diffusers else self.clip_preprocess(img)
        cross = self.clip(pre).float()[:, None]                                                     # [1, 1, 1024]
        if aug_noise is not None:
            noise = aug_noise.to(dev).reshape(img.shape)
        elif _diffusers:
            noise = torch.randn(img.shape, generator=self.gen, device=dev)                          # randn_tensor: NORMAL (diffusers pipeline, step 4)
        else:
            noise = torch.rand(img.shape, generator=self.gen, device=dev)                           # torch.rand_like: UNIFORM (streaming_svd.py:174)
        concat = self.enc(img + self.cond_aug * noise).float()                                      # [1, 4, h, w]
        vec = torch.cat([sinusoid([self.fps_id]), sinusoid([self.motion]), sinusoid([self.cond_aug])], -1).to(dev)   # [1, 768]
        c = dict(crossattn=cross.repeat(T, 1, 1), concat=concat.repeat(T, 1, 1, 1), vector=vec.repeat(T, 1))
        uc = dict(crossattn=torch.zeros_like(c["crossattn"]), concat=torch.zeros_like(c["concat"]), vector=c["vector"].clone())
        return c, uc
