"""StreamingSVD generation driver: one conditional chunk and the autoregressive outer loop.

Mirrors code/diffusion_trainer/streaming_svd.py:
  decode_first_stage              :123-151   (z / 0.18215, groups of 8 frames, fp32 output)
  _generate_conditional_output    :155-221   (noise -> EulerEDMSampler -> decode -> clamp)
  _autoregressive_generation      :293-356   (ctrl_frames = last 7 decoded frames, anchor = chunk0[6], keep result[7:])

Out of scope here: the conditioner (OpenCLIP ViT-H image tower + VAE encoder) that turns the anchor
frame into ``c``/``uc``.  It runs once per chunk; callers pass a ``conditioner(svd_input_frame) -> (c, uc)``
callable (the reference's GeneralConditioner, or synthetic tensors of the right shapes in bench/tests).
Noise is injected explicitly (``noise=``) because the reference draws from the global torch RNG
(streaming_svd.py:203), which is not reproducible across devices.
"""
import math

import torch

from . import ops
from .sampling import EulerEDMSampler


class StreamingSVD:
    def __init__(self, inference_model, first_stage_model, sampler=None, scale_factor=0.18215,
                 num_conditional_frames=7, use_memopt=False):
        self.inference_model = inference_model          # StreamingWrapper
        self.first_stage_model = first_stage_model      # object with .decode(z, timesteps=n) and .decoder
        self.sampler = sampler or EulerEDMSampler()
        self.scale_factor = scale_factor
        self.num_conditional_frames = num_conditional_frames
        self.use_memopt = use_memopt
        self.initial_num_steps = 25        # diffusers StableVideoDiffusionPipeline default num_inference_steps (streaming_svd.py:390)
        # CHUNK 0 runs on the STOCK SVD-XT weights in the reference (config.yaml:280-299 svd_pipeline = StableVideoDiffusionPipeline.from_pretrained(
        # "stabilityai/stable-video-diffusion-img2vid-xt")): pipeline.load_stock_svd_xt fills these three.  None = this checkpoint's own
        # `model.diffusion_model.*` / decoder / conditioner (whether StreamingSVD's training left them equal to the stock weights is not checkable offline).
        self.initial_model = None              # StreamingWrapper(VideoUNet(controlnet_mode=False) with the stock weights, None, Tc)
        self.initial_first_stage_model = None  # the stock AutoencoderKLTemporalDecoder's decoder
        self.initial_conditioner = None        # conditioner.SVDConditioner on the stock image_encoder / VAE encoder

    def set_initial_model(self, inference_model, first_stage_model=None, conditioner=None):
        self.initial_model, self.initial_first_stage_model, self.initial_conditioner = inference_model, first_stage_model, conditioner
        return self

    @torch.no_grad()
    def decode_first_stage(self, z, clamp=False, first_stage_model=None):
        """z / 0.18215, frames decoded in groups of 8 (4 with memopt) exactly like the reference (the temporal convolutions zero-pad at the
        group boundaries, so the grouping is part of the result).  The groups are independent: when the decoder carries a process group
        (`first_stage_model.decode_group`, set by parallel.JobPlan.attach) group n is decoded by rank n % world and broadcast -- every rank
        ends up with all frames (the next chunk's control frames), bit-identical to the single-process decode."""
        fsm = first_stage_model or self.first_stage_model
        z = z * (1.0 / self.scale_factor)
        n_samples = min(z.shape[0], 4 if self.use_memopt else 8)
        n_groups = math.ceil(z.shape[0] / n_samples)
        group = getattr(fsm, "decode_group", None)
        if group is None:
            outs = []
            for n in range(n_groups):
                zc = z[n * n_samples:(n + 1) * n_samples]
                outs.append(fsm.decode(zc, timesteps=len(zc), clamp=clamp))
            return torch.cat(outs, dim=0)
        import torch.distributed as dist
        from . import parallel
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        out = None
        mine = {}
        for n in range(rank, n_groups, world):
            zc = z[n * n_samples:(n + 1) * n_samples]
            mine[n] = fsm.decode(zc, timesteps=len(zc), clamp=clamp)
        shape = fsm.output_shape(z) if not mine else (z.shape[0],) + tuple(next(iter(mine.values())).shape[1:])
        out = torch.empty(shape, dtype=torch.float32, device=z.device)
        for n in range(n_groups):
            part = out[n * n_samples:(n + 1) * n_samples]
            if n in mine:
                part.copy_(mine[n])
            parallel.broadcast(part, src=dist.get_global_rank(group, n % world), group=group)
        return out

    @torch.no_grad()
    def _generate_conditional_output(self, c, uc, ctrl_frames, noise, num_steps=None):
        """c/uc: per-frame-repeated conditioning dicts ('crossattn' [T,1,1024], 'concat' [T,4,h,w], 'vector' [T,768]);
        ctrl_frames [1, Tc, 3, H, W] in [-1,1]; noise [T,4,h,w].  Returns frames [T,3,H,W] fp32 clamped to [-1,1]."""
        T = self.sampler.guider.num_frames
        assert noise.shape[0] == T
        x = noise.clone().float().contiguous()
        samples_z = self.sampler(self.inference_model, x, c, uc, num_steps=num_steps, batch_size=2,
                                 num_video_frames=T, ctrl_frames=ctrl_frames)
        return self.decode_first_stage(samples_z, clamp=True)          # torch.clamp(-1, 1) fused (streaming_svd.py:220)

    @torch.no_grad()
    def _generate_initial_chunk(self, c, uc, noise, num_steps=None, min_scale=1.0, max_scale=3.0):
        """Chunk 0.  The reference delegates the first 25 frames to diffusers' StableVideoDiffusionPipeline.__call__(image,
        decode_chunk_size=8) (streaming_svd.py:388-390); this follows that call (restated in diffusers' own formulation in
        oracle/svd_pipeline_oracle.py, checked against this method in tests/test_host_svd_cpu.py):
          * c / uc from `SVDConditioner.first_chunk` -- 0.02 * N(0, 1) noise augmentation, `_resize_with_antialiasing` for CLIP, un-scaled posterior
            mode, added_time_ids (6, 127, 0.02), zero negative branch (image_to_video below does that);
          * latents = noise * sqrt(sigma_max^2 + 1) (EulerDiscreteScheduler.init_noise_sigma, "leading" spacing), 25 Karras sigmas 700 -> 0.002 (rho 7),
            t = 0.25 ln sigma, v-prediction Euler steps -- term by term the EDM Euler step with VScaling of the AR chunks' sampler;
          * per-frame guidance linspace(1.0, 3.0, 25); the UNet WITHOUT ControlNet / CAM; decode in groups of 8 frames;
          * on the STOCK SVD-XT weights when they were loaded (`initial_model`, pipeline.load_stock_svd_xt), else on this checkpoint's own UNet / decoder.
        The third-party call itself is un-vendored and not installed: **parity unpinned** (SURVEY 8f N2)."""
        from .sampling import EDMDiscretization
        T = self.sampler.guider.num_frames
        num_steps = num_steps or self.initial_num_steps
        sampler = EulerEDMSampler(num_steps=num_steps, num_frames=T, min_scale=min_scale, max_scale=max_scale,
                                  discretization=EDMDiscretization(), cfg_exchange=self.sampler.cfg_exchange,
                                  use_graph=getattr(self.sampler, "use_graph", False))
        sampler._graph_pool = getattr(self.sampler, "_graph_pool", None)
        x = noise.clone().float().contiguous()
        z = sampler(self.initial_model or self.inference_model, x, c, uc, batch_size=2, num_video_frames=T, ctrl_frames=None)
        return self.decode_first_stage(z, clamp=True, first_stage_model=self.initial_first_stage_model)

    def initial_conditioning(self, conditioner, image):
        """(c, uc) of chunk 0: the stock pipeline's conditioner when loaded, diffusers' semantics (`first_chunk`) when the conditioner offers them."""
        cond0 = self.initial_conditioner or conditioner
        return cond0.first_chunk(image) if hasattr(cond0, "first_chunk") else cond0(image)

    @torch.no_grad()
    def image_to_video(self, conditioner, image, num_frames, noises, num_steps=None):
        """Mirror of StreamingSVD.image_to_video + inference_i2v.StreamingPipeline.image_to_video's chunk arithmetic
        (streaming_svd.py:359-402, inference_i2v.py:179-190): first chunk, then ceil((N - 25) / (25 - 7)) AR chunks,
        result cut to num_frames.  image [3, H, W] in [-1, 1]; conditioner(frame) -> (c, uc); noises: one [T,4,h,w] per chunk."""
        T, Tc = self.sampler.guider.num_frames, self.num_conditional_frames
        n_ar = max(0, math.ceil((num_frames - T) / (T - Tc)))
        assert len(noises) >= 1 + n_ar
        c, uc = self.initial_conditioning(conditioner, image)
        first = self.quantize_like_pil(self._generate_initial_chunk(c, uc, noises[0]))
        video = self._autoregressive_generation(first, conditioner, n_ar, noises[1:], num_steps=num_steps)
        return video[:num_frames]

    @staticmethod
    def quantize_like_pil(frames):
        """Chunk 0 reaches the autoregressive loop through PIL (streaming_svd.py:388-394): the diffusers pipeline returns uint8 frames
        ((x / 2 + 0.5).clamp(0, 1) * 255).round(), and `ToTensor()(frame) * 2.0 - 1` brings them back -- the first 25 frames, the anchor and
        the first control frames are therefore 8-bit values.  [-1, 1] fp32 in, [-1, 1] fp32 on the 1/255 grid out (format conversion)."""
        return ((frames / 2 + 0.5).clamp(0, 1) * 255.0).round() / 255.0 * 2.0 - 1

    @staticmethod
    def extract_ctrl_frames(video, num_conditional_frames):
        """Last frames of the previous chunk as [1, Tc, 3, H, W] (streaming_svd.py:263-290).  The reference first sends the chunk through
        convert_range([-1, 1] -> [-1, 1]) (utils/result_processor.py:4-14): (v + 1) / 2 * 2 - 1 in fp32 is not the identity in the last bit,
        and is reproduced here (7 frames, once per chunk) so that the hand-over is bit-identical."""
        v = video[-num_conditional_frames:][None]
        v = (v - (-1.0)) / 2.0
        return (v * 2.0 + (-1.0)).contiguous()

    @torch.no_grad()
    def _autoregressive_generation(self, initial_generation, conditioner, n_autoregressive_generations, noises,
                                   anchor_index=6, num_steps=None):
        """initial_generation [T0,3,H,W] in [-1,1] (chunk 0); returns all frames [T0 + n*(T-Tc), 3, H, W]."""
        Tc = self.num_conditional_frames
        result_chunks = [initial_generation]
        anchor = initial_generation[anchor_index]                         # streaming_svd.py:336
        for k in range(n_autoregressive_generations):
            ctrl_frames = self.extract_ctrl_frames(result_chunks[-1], Tc)
            c, uc = conditioner(anchor)
            result = self._generate_conditional_output(c, uc, ctrl_frames, noises[k], num_steps=num_steps)
            result_chunks.append(result[Tc:])                             # the overlap frames are re-generated, dropped
        return torch.cat(result_chunks, dim=0)

    @staticmethod
    def to_uint8_video(frames):
        """[-1, 1] fp32 frames [F, 3, H, W] -> uint8 [F, H, W, 3] exactly as the reference stores its result
        (convert_range at streaming_svd.py:353 + IImage/torch2np truncation, iimage.py:35-36)."""
        return ops.frames_to_uint8(frames.float().contiguous())
