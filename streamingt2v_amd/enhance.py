"""Enhancement stage: SDEdit denoising with the I2VGen-XL UNet and randomized blending (call stack §3.5).

Mirrors the denoising loop of the reference's ``I2VGenXLPipeline.__call__`` (code/i2v_enhance/pipeline_i2vgen_xl.py:812-913):

    scheduler.set_timesteps(30); timesteps = timesteps[t_start:]   (strength 0.97 -> 29 steps)            :812-816, 541-551
    latents = add_noise(video_latents, noise, timesteps[0])                                               :605-613
    for t in timesteps:
        for idx, window in chunks(latents, chunk_size, overlap):       # independent given `latents`      :846
            pred = unet(cat[window] * 2, t, text, fps, image_latents[idx], image_embeddings[idx])        :851-867
            pred = uncond + guidance_scale * (text - uncond)                                              :870-874
            window = scheduler.step(pred, t, window)                                                      :884-885
            latents_denoised[:, :, start + r : start + chunk] = window[:, :, r:]      r random            :891-903
        latents = latents_denoised

The UNet forward, the guidance and the DDIM update run in libsvdhip.so (i2vgen_unet.py, svd_ddim_cfg_step); the window
bookkeeping is blending.py (single device) or its rank-sharded form.  What is NOT here: the CLIP text /
image encoders and the 2-D AutoencoderKL that produce ``prompt_embeds``, ``image_embeddings``, ``image_latents`` and the
video latents -- the caller passes those tensors, exactly the arguments the reference's loop consumes.
"""
import random

import torch

from . import blending, ops


class DDIMSchedule:
    """The subset of diffusers==0.30.2 DDIMScheduler the loop uses (set_timesteps 'leading' + steps_offset, add_noise, step with
    eta 0), configured like ali-vilab/i2vgen-xl's scheduler_config.json as far as it is known offline: scaled_linear betas
    0.00085..0.012, 1000 train steps, v_prediction, rescale_betas_zero_snr, steps_offset 1, set_alpha_to_one False.  Host-side
    fp64 table math; the per-element update is the HIP kernel.  (diffusers is not vendored: parity unpinned, see oracle header.)"""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, prediction_type="v_prediction",
                 rescale_betas_zero_snr=True, set_alpha_to_one=False, beta_schedule="scaled_linear", timestep_spacing="leading"):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train, dtype=torch.float32)
        else:
            raise NotImplementedError(f"DDIM beta_schedule {beta_schedule!r}")
        if prediction_type not in ("v_prediction", "epsilon") or timestep_spacing not in ("leading", "trailing", "linspace"):
            raise NotImplementedError(f"DDIM prediction_type {prediction_type!r} / timestep_spacing {timestep_spacing!r}")
        self.spacing = timestep_spacing
        if rescale_betas_zero_snr:
            ac = torch.cumprod(1.0 - betas, 0).sqrt()
            a0, aT = ac[0].clone(), ac[-1].clone()
            ac = (ac - aT) * a0 / (a0 - aT)
            ab = ac ** 2
            betas = 1 - torch.cat([ab[0:1], ab[1:] / ab[:-1]])
        self.alphas_cumprod = torch.cumprod(1.0 - betas, 0)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.num_train, self.offset, self.v_prediction = num_train, steps_offset, prediction_type == "v_prediction"

    @classmethod
    def from_config(cls, cfg):
        """scheduler/scheduler_config.json of a diffusers pipeline folder (what DDIMScheduler.from_pretrained reads).  Options that would
        change the update rule and are not implemented (sample clipping, thresholding) are refused instead of ignored."""
        if cfg.get("clip_sample", False) or cfg.get("thresholding", False):
            raise NotImplementedError("DDIM clip_sample / thresholding are not implemented (ali-vilab/i2vgen-xl sets both to false)")
        return cls(num_train=cfg.get("num_train_timesteps", 1000), beta_start=cfg.get("beta_start", 0.0001), beta_end=cfg.get("beta_end", 0.02),
                   steps_offset=cfg.get("steps_offset", 0), prediction_type=cfg.get("prediction_type", "epsilon"),
                   rescale_betas_zero_snr=cfg.get("rescale_betas_zero_snr", False), set_alpha_to_one=cfg.get("set_alpha_to_one", True),
                   beta_schedule=cfg.get("beta_schedule", "linear"), timestep_spacing=cfg.get("timestep_spacing", "leading"))

    def set_timesteps(self, n):
        """diffusers DDIMScheduler.set_timesteps for the three spacings."""
        import numpy as np
        self.n, T = n, self.num_train
        if self.spacing == "leading":
            ts = (np.arange(0, n) * (T // n)).round()[::-1].astype(np.int64) + self.offset
        elif self.spacing == "trailing":
            ts = np.round(np.arange(T, 0, -T / n)).astype(np.int64) - 1
        else:
            ts = np.linspace(0, T - 1, n).round()[::-1].astype(np.int64)
        self.timesteps = [int(t) for t in ts]
        return self.timesteps

    def get_timesteps(self, n, strength):
        """pipeline_i2vgen_xl.py:541-551 (img2img): drop the first n - int(n * strength) steps."""
        self.set_timesteps(n)
        t_start = max(n - min(int(n * strength), n), 0)
        return self.timesteps[t_start:]

    def alphas(self, t):
        prev = t - self.num_train // self.n
        return float(self.alphas_cumprod[t]), (float(self.alphas_cumprod[prev]) if prev >= 0 else self.final_alpha_cumprod)

    def add_noise(self, x0, noise, t):
        a = float(self.alphas_cumprod[t])
        return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


class I2VEnhancer:
    """unet: streamingt2v_amd.i2vgen_unet.I2VGenXLUNet (weights loaded).  One instance per process / GPU."""

    def __init__(self, unet, scheduler=None, guidance_scale=9.0, num_inference_steps=30, strength=0.97):
        self.unet, self.sched = unet, scheduler or DDIMSchedule()
        self.g, self.steps, self.strength = guidance_scale, num_inference_steps, strength
        self._consts = {}

    def _denoise_window(self, window, t, cond):
        """window fp32 [1, 4, chunk, h, w] -> DDIM-updated window.  cond = dict(fps [2], image_latents [2,4,chunk,h,w],
        image_embeddings [2,cd], text [2,77,cd]) with the unconditional half first (pipeline order, :770-796)."""
        _, C, Fr, H, W = window.shape
        const = self._consts.get(id(cond))
        if const is None:        # once per window: constant over the DDIM steps (the reference recomputes it in every forward)
            const = self._consts[id(cond)] = self.unet.set_conditioning(cond["fps"], cond["image_latents"], cond["image_embeddings"],
                                                                      cond["text"])
        self.unet.use_conditioning(const)
        fr = window[0].permute(1, 0, 2, 3).contiguous()                      # [chunk, 4, h, w]
        pred = self.unet.forward_frames(torch.cat([fr, fr], 0), t)            # [(2 chunk), 4, h, w]: uncond | text
        a_t, a_prev = self.sched.alphas(t)
        out = ops.ddim_cfg_step(fr, pred[:Fr].contiguous(), pred[Fr:].contiguous(), self.g, a_t, a_prev, self.sched.v_prediction)
        return out.permute(1, 0, 2, 3)[None]

    # ---- (window, CFG half) units: what a rank evaluates when the step is sharded over a process group (blending.blend_step_units_sharded) ----
    def _predict_half(self, window, t, cond, half):
        """Raw UNet prediction [chunk, 4, h, w] of ONE CFG half (0 = unconditional, 1 = text) of a window: the batch-1 evaluation with that
        half's conditioning.  Equal to its half of the batched evaluation of _denoise_window (per-sample norms and attention)."""
        key = (id(cond), half)
        const = self._consts.get(key)
        if const is None:
            sl = slice(half, half + 1)
            const = self._consts[key] = self.unet.set_conditioning(cond["fps"][sl], cond["image_latents"][sl], cond["image_embeddings"][sl],
                                                                 cond["text"][sl])
        self.unet.use_conditioning(const)
        fr = window[0].permute(1, 0, 2, 3).contiguous()
        return self.unet.forward_frames(fr, t)

    def _combine_halves(self, window, t, pred_uncond, pred_text):
        """guidance (pipeline_i2vgen_xl.py:870-874) + DDIMScheduler.step (:884-885) on a window, from the two halves' predictions."""
        fr = window[0].permute(1, 0, 2, 3).contiguous()
        a_t, a_prev = self.sched.alphas(t)
        out = ops.ddim_cfg_step(fr, pred_uncond.contiguous(), pred_text.contiguous(), self.g, a_t, a_prev, self.sched.v_prediction)
        return out.permute(1, 0, 2, 3)[None]

    def _denoise_window_pair(self, window, t, cond, cfg_exchange):
        """One window on a CFG pair of ranks (x sequence parallelism inside each half when `unet.sp` is set): this rank evaluates ITS half
        (even rank of the pair: unconditional, odd: text), the pair all-gathers the two predictions (8.75 MB each for a 38-frame window) and
        both apply guidance + DDIM -- every rank ends with the same window."""
        Fr = window.shape[2]
        both = cfg_exchange.gather(self._predict_half(window, t, cond, cfg_exchange.half).contiguous())
        return self._combine_halves(window, t, both[:Fr], both[Fr:])

    def denoise(self, video_latents, noise, conds, chunk_size, overlap_size, rng=random, group=None, shard="units", plan=None):
        """video_latents / noise fp32 [1, 4, F, h, w]; conds: one dict per window.  Returns the enhanced latents.
        group: a process group -> every DDIM step is sharded over its ranks; shard = "units" (default): the 2 x n_windows (window, CFG half)
        units round-robin (3 windows keep 6 of 8 GPUs busy), "windows": whole windows (the round-1 form: 3 of 8).
        plan (round 5): a parallel.JobPlan in "job" mode -- the SAME partition as stage 1: CFG pair x frame <-> pixel sequence parallelism of
        degree world / 2 inside each half (I2VGenXLUNet.forward_frames with `sp`); the windows of a step run one after the other, each on ALL
        ranks, so 8 GPUs are 8 busy GPUs whatever the number of windows (units: 6 of 8 for the shipped 3-window job)."""
        ts = self.sched.get_timesteps(self.steps, self.strength)
        self._consts = {}
        latents = self.sched.add_noise(video_latents, noise, ts[0])
        n_chunks = len(conds)
        if plan is not None and plan.cfg_exchange is not None:
            # a window with fewer frames than the sequence-parallel degree (the 3-frame key-frame pre-pass) runs on the CFG pair alone: the ranks of a
            # half then compute the same prediction redundantly (identical results, no collective inside the half)
            sp = plan.sp if (plan.sp is not None and min(chunk_size, video_latents.shape[2]) >= plan.sp.size) else None
            prev_sp, self.unet.sp = self.unet.sp, sp
            try:
                for t in ts:
                    fn = lambda idx, w, t=t: self._denoise_window_pair(w, t, conds[idx], plan.cfg_exchange)
                    latents = blending.blend_step(latents, fn, chunk_size, overlap_size, n_chunks, rng)
            finally:
                self.unet.sp = prev_sp
            return latents
        for t in ts:
            fn = lambda idx, w, t=t: self._denoise_window(w, t, conds[idx])
            if group is None:
                latents = blending.blend_step(latents, fn, chunk_size, overlap_size, n_chunks, rng)
            elif shard == "windows":
                latents = blending.blend_step_sharded(latents, fn, chunk_size, overlap_size, n_chunks, rng, group)
            else:
                ph = lambda idx, half, w, t=t: self._predict_half(w, t, conds[idx], half)
                cb = lambda idx, w, pu, pc, t=t: self._combine_halves(w, t, pu, pc)
                latents = blending.blend_step_units_sharded(latents, ph, cb, chunk_size, overlap_size, n_chunks, rng, group)
        return latents
