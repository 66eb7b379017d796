"""Stage-1 conditioner on the MI355X kernels: what `GeneralConditioner` produces for one conditioning frame in
StreamingSVD._generate_conditional_output (code/diffusion_trainer/streaming_svd.py:166-199, embedders of config.yaml:160-218):

    crossattn [T, 1, 1024]  OpenCLIP ViT-H/14 image embedding of the frame            (clip_vision.OpenCLIPVisionTower)
    concat    [T, 4, h, w]  AutoencoderKLModeOnly.encode(frame + cond_aug * U[0,1) noise), scale 1   (temporal_ae.CondFrameEncoder)
    vector    [T, 768]      sinusoidal embeddings (256 each) of fps_id 6, motion_bucket_id 127, cond_aug 0.02, in that order
    uc        crossattn and concat zeroed (force_uc_zero_embeddings), vector unchanged

The only piece that is not the reference's arithmetic: the 224 x 224 CLIP resize.  The reference uses kornia's antialiased bicubic
resize (modules.py:624-636; kornia is not vendored); here it is torch's antialiased bicubic interpolate -- close, not identical.
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


class SVDConditioner:
    def __init__(self, clip_tower, cond_encoder, num_frames=25, fps_id=6, motion_bucket_id=127, cond_aug=0.02, generator=None):
        self.clip, self.enc, self.T = clip_tower, cond_encoder, num_frames
        self.fps_id, self.motion, self.cond_aug, self.gen = fps_id, motion_bucket_id, cond_aug, generator

    @staticmethod
    def clip_preprocess(img):
        """[n, 3, H, W] in [-1, 1] -> CLIP-normalised [n, 3, 224, 224] (modules.py:624-636)."""
        x = F.interpolate(img.float(), size=(224, 224), mode="bicubic", align_corners=True, antialias=True)
        x = (x + 1.0) / 2.0
        mean = torch.tensor(CLIP_MEAN, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    @torch.no_grad()
    def __call__(self, frame):
        """frame [3, H, W] fp32 in [-1, 1] (on the GPU) -> (c, uc) dicts as StreamingSVD._generate_conditional_output consumes them."""
        T = self.T
        img = frame[None].float()
        dev = img.device
        cross = self.clip(self.clip_preprocess(img)).float()[:, None]                               # [1, 1, 1024]
        noise = torch.rand(img.shape, generator=self.gen, device=dev)                               # torch.rand_like: UNIFORM (streaming_svd.py:174)
        concat = self.enc(img + self.cond_aug * noise).float()                                      # [1, 4, h, w]
        vec = torch.cat([sinusoid([self.fps_id]), sinusoid([self.motion]), sinusoid([self.cond_aug])], -1).to(dev)   # [1, 768]
        c = dict(crossattn=cross.repeat(T, 1, 1), concat=concat.repeat(T, 1, 1, 1), vector=vec.repeat(T, 1))
        uc = dict(crossattn=torch.zeros_like(c["crossattn"]), concat=torch.zeros_like(c["concat"]), vector=c["vector"].clone())
        return c, uc
