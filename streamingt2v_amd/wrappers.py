"""StreamingWrapper: the drop-in boundary of the denoising hot path.

Mirrors code/models/diffusion/wrappers.py:7-78 -- same constructor arguments, same
``forward(x, t, c, *, batch_size, num_video_frames, image_only_indicator, ctrl_frames)`` contract, NCHW fp32
tensors in and out -- so it can take the place of ``model`` in ``Denoiser.forward`` (denoiser.py:36-38).
"""
import torch

from . import ops


class StreamingWrapper:
    def __init__(self, diffusion_model, controlnet, num_frame_conditioning, compile_model=False,
                 pipeline_offloading=False):
        if pipeline_offloading:
            raise NotImplementedError("Pipeline offloading for StreamingI2V not implemented yet.")   # wrappers.py:19-21
        self.diffusion_model = diffusion_model
        self.controlnet = controlnet
        self.num_frame_conditioning = num_frame_conditioning

    def requires_grad_(self, flag=False):
        return self

    def eval(self):
        return self

    # ---- shared core on token tensors ------------------------------------------------------------------------
    sp = None      # parallel.SeqParallel: the forward runs frame <-> pixel sequence-parallel over its group (set by parallel.JobPlan.attach)

    def _run(self, x_tok, t, context, y, batch_size, T, H, W, ctrl_frames):
        """x_tok [B*T*pix, 32] tokens of ALL frames (cheap); with self.sp every rank slices its frames out of them, runs its share of the
        ControlNet and of the UNet, and the UNet all-gathers the output: the result is the same [B*T*pix, 4] fp32 on every rank."""
        Tc = self.num_frame_conditioning
        sp, pix = self.sp, H * W
        hs_c = mid_c = None
        # no control frames (first chunk: plain SVD, video_model.py:582,603 with hs_control_* = None) -> UNet only
        if self.diffusion_model.controlnet_mode and ctrl_frames is not None:
            # ControlNet sees the first Tc frames of each CFG half (wrappers.py:28-42) ...
            def reduce_rows(v, per):           # "(B F) ... -> B F ..." [:, :Tc]
                v = v.reshape(batch_size, T * per, *v.shape[1:])[:, :Tc * per]
                return v.reshape(batch_size * Tc * per, *v.shape[2:]).contiguous()

            x_ctrl = reduce_rows(x_tok, pix)
            t_ctrl = reduce_rows(t, 1)
            # only CLIP token 0 (wrappers.py:39).  One derived tensor per `context` OBJECT: the networks recognise a chunk's constant context by
            # identity and compute its cross-attention constants once per chunk instead of once per Euler step (video_model._attn2_const)
            held = getattr(self, "_ctx_ctrl", None)
            ver = ops.tensor_version(context)
            if held is None or ver is None or held[0] is not context or held[1] != ver or held[2] != (batch_size, T, Tc):
                self._ctx_ctrl = held = (context, ver, (batch_size, T, Tc), reduce_rows(context[:, :1], 1))
            ctx_ctrl = held[3]
            y_ctrl = reduce_rows(y, 1)
            # ... and the pixel-space control frames, repeated for both CFG halves (wrappers.py:45-48)
            # "(2 B) F ..." in the reference (B = 1 video, CFG batch 2); one copy per batch element here so that a rank
            # holding a single CFG half (parallel.CfgPairExchange) repeats once
            cond = self._cond_cached(ctrl_frames, batch_size)
            assert self.controlnet.stem_x3 == self.diffusion_model.stem_x3, "UNet and ControlNet were loaded under different precision plans (ops.EXACT_RIM)"
            if sp is not None:
                assert sp.size <= Tc, "sequence parallelism shards the Tc conditioning frames of the ControlNet: degree <= Tc"
                x_ctrl = sp.take_frames(x_ctrl, batch_size, Tc, pix)
            hs_c, mid_c = self.controlnet.forward_tokens(x_ctrl, t_ctrl, cond, ctx_ctrl, y_ctrl, Tc, H, W, sp=sp)
        if sp is not None:
            x_tok = sp.take_frames(x_tok, batch_size, T, pix)
        return self.diffusion_model.forward_tokens(x_tok, t, context, y, T, H, W, hs_c, mid_c, Tc, sp=sp)

    def _cond_cached(self, ctrl_frames, batch_size):
        # ONE repeated fp32 tensor per ctrl_frames OBJECT (built on a miss only) so ControlNet.embed_condition can recognise it.  The
        # cache holds a reference to ctrl_frames itself and compares identity + in-place version: a (data_ptr, shape, version) key can
        # collide once the tensor is freed and the caching allocator hands the same address to the next video's control frames.
        # With sequence parallelism the cached tensor holds this rank's share of the conditioning frames.
        held = getattr(self, "_cond_src", None)
        ver = ops.tensor_version(ctrl_frames)
        if held is None or ver is None or held[0] is not ctrl_frames or held[1] != ver or held[2] != batch_size or held[3] is not self.sp:
            cond = ctrl_frames.repeat(batch_size // ctrl_frames.shape[0], *([1] * (ctrl_frames.dim() - 1))).flatten(0, 1)
            if self.sp is not None:
                cond = self.sp.take_frames(cond, batch_size, self.num_frame_conditioning)
            self._cond_src = (ctrl_frames, ver, batch_size, self.sp)
            self._cond_val = cond.float().contiguous()
        return self._cond_val

    def reset_caches(self):
        """Release the per-chunk caches of the wrapper and of both networks (see _EncoderBase.reset_caches): between videos."""
        self._ctx_ctrl = self._cond_src = self._cond_val = None
        for net in (self.diffusion_model, self.controlnet):
            if net is not None and hasattr(net, "reset_caches"):
                net.reset_caches()

    # ---- reference-shaped entry point ------------------------------------------------------------------------
    def forward(self, x, t, c, **kwargs):
        batch_size = kwargs.pop("batch_size")
        T = kwargs.pop("num_video_frames")
        ioi = kwargs.pop("image_only_indicator", None)
        if ioi is not None and bool(ioi.any()):
            raise NotImplementedError("image_only_indicator != 0 is unused by StreamingSVD (streaming_svd.py:206)")
        ctrl_frames = kwargs.get("ctrl_frames")
        F, _, H, W = x.shape
        concat = c.get("concat")
        x_tok = self.diffusion_model.input_tokens(x.float().contiguous(), concat.float().contiguous() if concat is not None else None, None)
        out = self._run(x_tok, t.float().contiguous(), c["crossattn"].float(), c["vector"].float().contiguous(),
                        batch_size, T, H, W, ctrl_frames)
        return ops.tokens_to_nchw(out, self.diffusion_model.out_channels, F, H, W)

    __call__ = forward

    # ---- fused entry point used by our EulerEDMSampler ---------------------------------------------------------
    def forward_fused(self, x, c_in, c_noise, c2, *, batch_size, num_video_frames, ctrl_frames=None,
                      image_only_indicator=None, **_ignored):
        """x [T,4,h,w] fp32 (UNscaled sampler state); c2: cond dict with batch_size*T rows (batch_size = 2: CFG-doubled
        (uncond | cond); batch_size = 1: ONE CFG half, used by the CFG-pair split over two ranks).  Returns the raw network
        output tokens [batch_size*T*h*w, 4] fp32, i.e. network(cat([x]*batch_size) * c_in, c_noise, c2)."""
        scale, tvec = self.step_scalars(x, batch_size)
        scale.fill_(c_in)
        tvec.fill_(c_noise)
        return self.forward_fused_static(x, c2, batch_size=batch_size, num_video_frames=num_video_frames, ctrl_frames=ctrl_frames)

    def step_scalars(self, x, batch_size):
        """The two per-step scalars of a fused forward (c_in, c_noise; denoiser_scaling.py:51-59) as DEVICE vectors [batch_size * T]: the only
        inputs of `forward_fused_static` that change between the Euler steps of a chunk besides x itself -- which is what makes one captured
        hipGraph of the forward replayable for every step (sampling.EulerEDMSampler(use_graph=True))."""
        T = x.shape[0]
        key = (x.device, T, batch_size)
        aux = getattr(self, "_aux", {}).get(key)
        if aux is None:
            aux = (torch.empty((batch_size * T,), dtype=torch.float32, device=x.device),
                   torch.empty((batch_size * T,), dtype=torch.float32, device=x.device))
            self._aux = getattr(self, "_aux", {})
            self._aux[key] = aux
        return aux

    def forward_fused_static(self, x, c2, *, batch_size, num_video_frames, ctrl_frames=None):
        """forward_fused with c_in / c_noise read from the `step_scalars` vectors: no host value enters a kernel argument, no allocation
        outlives the call except the returned tensor, nothing synchronises -- capturable in a hipGraph."""
        T, _, H, W = x.shape
        scale, tvec = self.step_scalars(x, batch_size)
        x2 = torch.cat([x] * batch_size, 0) if batch_size > 1 else x
        x_tok = self.diffusion_model.input_tokens(x2, c2["concat"], scale)          # (x * c_in | concat) -> 8 ch, padded to 32 (split-3 for a rim stem)
        return self._run(x_tok, tvec, c2["crossattn"], c2["vector"], batch_size, num_video_frames, H, W, ctrl_frames)
