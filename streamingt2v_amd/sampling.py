"""Euler-EDM sampling stack of StreamingSVD (host loop in Python, arithmetic in libsvdhip.so kernels).

Mirrors:
  AlignYourSteps                      <- code/models/diffusion/discretizer.py:8-33 (+ append_zero, sgm discretizer.py:18-22)
  VScalingWithEDMcNoise               <- code/models/svd/sgm/modules/diffusionmodules/denoiser_scaling.py:51-59
  Denoiser (fused: svd_edm_euler_step) <- .../denoiser.py:11-39
  LinearPredictionGuider              <- .../guiders.py:60-99
  EulerEDMSampler (s_churn = 0)       <- .../sampling.py:41-52, 93-130, 211-215

The sigma schedule is float64 on the host exactly as in the reference; per-step sigmas reach the kernels as fp32.
One fused way to run a step (``EulerEDMSampler.__call__`` with a StreamingWrapper): the c_in scaling rides in the NCHW->token conversion
kernel, and denoiser-combine + guidance + Euler update are one kernel on the fp32 state.  The reference-shaped route -- the reference's OWN
``EulerEDMSampler`` / ``Denoiser`` objects calling ``StreamingWrapper.forward(x * c_in, c_noise, cond, **kw)`` -- needs nothing from this
module: the wrapper keeps the reference's forward contract.
"""
import math

import numpy as np
import torch

from . import ops


class AlignYourSteps:
    """Log-linear interpolation of the 11-point AYS schedule (sigma_max 700)."""

    SCHEDULE = [700.00, 54.5, 15.886, 7.977, 4.248, 1.789, 0.981, 0.403, 0.173, 0.034, 0.002]

    def __init__(self, sigma_min=0.002, sigma_max=700.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n):
        t = np.array(self.SCHEDULE)
        xs = np.linspace(0, 1, len(t))
        ys = np.log(t[::-1])
        new_ys = np.interp(np.linspace(0, 1, n), xs, ys)
        return np.exp(new_ys)[::-1].copy()

    def __call__(self, n, do_append_zero=True):
        s = self.get_sigmas(n)
        return np.concatenate([s, [0.0]]) if do_append_zero else s


class EDMDiscretization:
    """Karras rho-schedule: (smax^(1/rho) + ramp * (smin^(1/rho) - smax^(1/rho)))^rho.
    sgm discretizer.py:27-38; identical to diffusers' EulerDiscreteScheduler(use_karras_sigmas) that the reference's first
    chunk runs through StableVideoDiffusionPipeline (sigma_min 0.002, sigma_max 700, 25 steps; streaming_svd.py:388-390)."""

    def __init__(self, sigma_min=0.002, sigma_max=700.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n):
        ramp = np.linspace(0, 1, n, dtype=np.float32)           # the reference builds the ramp with torch.linspace (fp32)
        mn, mx = self.sigma_min ** (1 / self.rho), self.sigma_max ** (1 / self.rho)
        return ((mx + ramp * (mn - mx)) ** self.rho).astype(np.float64)

    def __call__(self, n, do_append_zero=True):
        s = self.get_sigmas(n)
        return np.concatenate([s, [0.0]]) if do_append_zero else s


class VScalingWithEDMcNoise:
    def __call__(self, sigma):
        s2 = sigma * sigma + 1.0
        return 1.0 / s2, -sigma / math.sqrt(s2), 1.0 / math.sqrt(s2), 0.25 * math.log(sigma)


class LinearPredictionGuider:
    def __init__(self, max_scale=3.0, num_frames=25, min_scale=1.5):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames)

    def prepare_inputs(self, x, s, c, uc):
        c_out = {k: torch.cat((uc[k], c[k]), 0) for k in ("vector", "crossattn", "concat")}
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class EulerEDMSampler:
    """30 Euler steps on the AYS schedule with per-frame linear guidance (config.yaml:139-158)."""

    def __init__(self, num_steps=30, num_frames=25, min_scale=1.5, max_scale=3.0, discretization=None, cfg_exchange=None, use_graph=False):
        """cfg_exchange: optional streamingt2v_amd.parallel.CfgPairExchange -- this rank then evaluates only ITS half
        of the CFG batch per step and all-gathers the raw network outputs with its partner (exact 2-way split).
        use_graph: capture the network evaluation of a chunk's SECOND step in a hipGraph (torch.cuda.CUDAGraph around the libsvdhip launches: they
        go to torch's current stream, which is the capture stream) and replay it for the remaining steps -- ~1 000 Python launcher calls per
        forward (26 us of host time each, profiles/r03_host_probe.txt) become one graph launch.  The shapes, the conditioning tensors and every
        cached constant are fixed within a chunk; the per-step scalars live in device vectors (StreamingWrapper.step_scalars).  Not used when the
        wrapper runs sequence-parallel (collectives inside the forward)."""
        self.cfg_exchange = cfg_exchange
        self.use_graph = use_graph
        self._graph_pool = None
        self.num_steps = num_steps
        self.discretization = discretization or AlignYourSteps()
        self.guider = LinearPredictionGuider(max_scale=max_scale, num_frames=num_frames, min_scale=min_scale)
        self.scaling = VScalingWithEDMcNoise()
        self._gscale = {}

    def sigmas(self, num_steps=None):
        return self.discretization(self.num_steps if num_steps is None else num_steps)

    def __call__(self, wrapper, x, cond, uc, num_steps=None, **model_kwargs):
        """x: [T, 4, h, w] fp32 noise (modified in place, like the reference's ``x *= sqrt(1 + sigma0^2)``);
        cond/uc: dicts with 'concat' [T,4,h,w], 'crossattn' [T,1,1024], 'vector' [T,768];
        wrapper: streamingt2v_amd.wrappers.StreamingWrapper.  Returns the denoised latents [T,4,h,w] fp32."""
        sig = self.sigmas(num_steps)                         # float64
        T = x.shape[0]
        assert T == self.guider.num_frames
        x.mul_(float(np.sqrt(1.0 + sig[0] ** 2.0)))           # prepare_sampling_loop (sampling.py:47)
        dev = x.device
        g = self._gscale.get(dev)
        if g is None:
            g = self.guider.scale.to(dev).float().contiguous()
            self._gscale[dev] = g
        ex = self.cfg_exchange
        if ex is None:
            c2 = {k: torch.cat((uc[k], cond[k]), 0).float().contiguous() for k in ("vector", "crossattn", "concat")}
        else:
            half = (uc, cond)[ex.half]
            c2 = {k: half[k].float().contiguous() for k in ("vector", "crossattn", "concat")}
            model_kwargs = dict(model_kwargs, batch_size=1)
        graph = graph_net = None
        # not under a launch trace / the in-situ tuner: both record timing-enabled events around every launch, which a stream capture refuses
        # (and a replayed step would emit no records at all)
        graphable = (self.use_graph and x.is_cuda and getattr(wrapper, "sp", None) is None and hasattr(wrapper, "forward_fused_static")
                     and ops.trace is None and ops.tuner is None
                     and set(model_kwargs) <= {"batch_size", "num_video_frames", "ctrl_frames", "image_only_indicator"})
        for i in range(len(sig) - 1):
            s = float(np.float32(sig[i]))                     # s_in * sigmas[i] is fp32 in the reference
            s_next = float(np.float32(sig[i + 1]))
            _, _, c_in, c_noise = self.scaling(s)
            if graph is not None:
                scale, tvec = wrapper.step_scalars(x, model_kwargs["batch_size"])
                scale.fill_(c_in)
                tvec.fill_(c_noise)
                graph.replay()
                net = graph_net
            else:
                net = wrapper.forward_fused(x, c_in, c_noise, c2, **model_kwargs)   # [2T*pix, 4] fp32 tokens
                if graphable and i == 0 and len(sig) > 3:
                    # step 0 ran eagerly (it also fills the per-chunk caches and the kernels' one-time attributes); capture the forward once
                    # now -- nothing executes during capture -- and replay it from step 1 on
                    graph, graph_net = self._capture(wrapper, x, c2, model_kwargs)
            if ex is not None:
                net = ex.gather(net)                          # (uncond | cond) from the two ranks of the pair
            ops.edm_euler_step(x, net, g, s, s_next)
        return x

    def _capture(self, wrapper, x, c2, model_kwargs):
        kw = {k: model_kwargs[k] for k in ("batch_size", "num_video_frames") if k in model_kwargs}
        kw["ctrl_frames"] = model_kwargs.get("ctrl_frames")
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()      # one private pool for all chunks' graphs (a chunk's graph replaces the last)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._graph_pool):
            net = wrapper.forward_fused_static(x, c2, **kw)
        return g, net
