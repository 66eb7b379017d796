"""Temporal VAE decoder (SVD's VideoDecoder, time_mode "conv-only") on the libsvdhip.so kernels.

Mirrors code/models/svd/sgm/modules/autoencoding/temporal_ae.py:16-105,291-347 and
code/models/svd/sgm/modules/diffusionmodules/model.py:94-201,603-748 (Decoder, ResnetBlock, AttnBlock, Upsample),
state_dict keys as under ``first_stage_model.decoder.`` of the reference checkpoint.

Same execution plan as the UNet: channels-last bf16 tokens, 3x3 convs and (3,1,1) temporal convs as implicit
GEMMs (nearest-2x upsample folded into the conv addressing), GroupNorm+SiLU kernels, bias/residual/alpha-blend
epilogues.  The single-head (d = 512) mid attention is QK^T GEMM (fp32 scores) -> row softmax -> PV GEMM per
frame.  The reference runs this decoder in fp32 (config.yaml:310); bf16 storage / fp32 accumulation here, the
parity tolerance is stated in tests/test_gpu_parity.py.
"""
import torch

from . import ops
from .params import Spec, check_state_dict
from .video_model import _Conv, _dev_bf16, _dev_f32, _sigmoid, pack_conv3x3, pack_tconv3, pad_rows


class VaeConfig:
    """decoder_config of config.yaml:241-258."""

    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, out_ch=3):
        self.ch, self.ch_mult, self.num_res_blocks = ch, tuple(ch_mult), num_res_blocks
        self.z_channels, self.out_ch = z_channels, out_ch


class AEVideoResBlock:
    """ResnetBlock (temb=None) -> time_stack ResBlock(3,1,1; no emb) -> alpha*temporal + (1-alpha)*spatial."""

    def __init__(self, prefix, cin, cout, stream=False):
        self.p, self.cin, self.cout = prefix, cin, cout
        self.stream = stream          # may carry the fp32 residual stream (the temporal VideoDecoder's blocks)

    def spec(self, s):
        p, ci, co = self.p, self.cin, self.cout
        s.add(p + "norm1.weight", ci); s.add(p + "norm1.bias", ci)
        s.add(p + "conv1.weight", co, ci, 3, 3); s.add(p + "conv1.bias", co)
        s.add(p + "norm2.weight", co); s.add(p + "norm2.bias", co)
        s.add(p + "conv2.weight", co, co, 3, 3); s.add(p + "conv2.bias", co)
        if ci != co:
            s.add(p + "nin_shortcut.weight", co, ci, 1, 1); s.add(p + "nin_shortcut.bias", co)
        t = p + "time_stack."
        s.add(t + "in_layers.0.weight", co); s.add(t + "in_layers.0.bias", co)
        s.add(t + "in_layers.2.weight", co, co, 3, 1, 1); s.add(t + "in_layers.2.bias", co)
        s.add(t + "out_layers.0.weight", co); s.add(t + "out_layers.0.bias", co)
        s.add(t + "out_layers.3.weight", co, co, 3, 1, 1); s.add(t + "out_layers.3.bias", co)
        s.add(p + "mix_factor", 1)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        Fv = lambda k: _dev_f32(g(k), dev)
        self.n1, self.n2 = (Fv("norm1.weight"), Fv("norm1.bias")), (Fv("norm2.weight"), Fv("norm2.bias"))
        self.w1, self.b1 = _dev_bf16(pack_conv3x3(g("conv1.weight")), dev), Fv("conv1.bias")
        self.w2, self.b2 = _dev_bf16(pack_conv3x3(g("conv2.weight")), dev), Fv("conv2.bias")
        if self.cin != self.cout:
            self.ws, self.bs = _dev_bf16(g("nin_shortcut.weight")[:, :, 0, 0], dev), Fv("nin_shortcut.bias")
        t = "time_stack."
        self.tn1, self.tn2 = (Fv(t + "in_layers.0.weight"), Fv(t + "in_layers.0.bias")), (Fv(t + "out_layers.0.weight"), Fv(t + "out_layers.0.bias"))
        self.tw1, self.tb1 = _dev_bf16(pack_tconv3(g(t + "in_layers.2.weight")), dev), Fv(t + "in_layers.2.bias")
        self.tw2, self.tb2 = _dev_bf16(pack_tconv3(g(t + "out_layers.3.weight")), dev), Fv(t + "out_layers.3.bias")
        self.alpha = _sigmoid(g("mix_factor"))

    def forward(self, x, F, H, W):
        pix = H * W
        cvi = dict(cin=self.cin, hin=H, win=W, hout=H, wout=W, frames=F)
        cv = dict(cin=self.cout, hin=H, win=W, hout=H, wout=W, frames=F)
        tv = dict(cin=self.cout, T=F, pix=pix)
        # st: this block's sums (h + skip, the alpha blend) are the decoder's residual stream: fp32 between the kernels under the decoder's precision plan
        # (ops.AE_STREAM_F32_MIN_CH); every convolution operand stays 16 bit.  The block's input is whatever the layer before wrote.
        st = self.stream and ops.ae_stream_on(self.cout)
        h = ops.groupnorm(x, F, pix, *self.n1, 1e-6, silu=True)
        h = ops.gemm(h, self.w1, bias=self.b1, conv=cvi)
        h = ops.groupnorm(h, F, pix, *self.n2, 1e-6, silu=True)
        if self.cin == self.cout:
            skip = x if (st or x.dtype != torch.float32) else ops.to_elem_rows(x)
        else:
            skip = ops.gemm(ops.to_elem_rows(x), self.ws, bias=self.bs, out_f32=st)
        hs = ops.gemm(h, self.w2, bias=self.b2, residual=skip, conv=cv, out_f32=st)
        g = ops.groupnorm(hs, F, pix, *self.tn1, 1e-5, frames_per_stat=F, silu=True)
        g = ops.gemm(g, self.tw1, bias=self.tb1, temporal=tv)
        g = ops.groupnorm(g, F, pix, *self.tn2, 1e-5, frames_per_stat=F, silu=True)
        # x = alpha * temporal + (1 - alpha) * spatial   (temporal_ae.py:77-78: opposite convention to the UNet)
        return ops.gemm(g, self.tw2, bias=self.tb2, residual=hs, blend=(1.0 - self.alpha, hs), temporal=tv, out_f32=st)


class AEAttnBlock:
    """AttnBlock (model.py:161-201): one head of width C over H*W tokens per frame."""

    def __init__(self, prefix, ch):
        self.p, self.c = prefix, ch
        self._vt = {}

    def spec(self, s):
        p, c = self.p, self.c
        s.add(p + "norm.weight", c); s.add(p + "norm.bias", c)
        for n in ("q", "k", "v", "proj_out"):
            s.add(p + n + ".weight", c, c, 1, 1); s.add(p + n + ".bias", c)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        self.dev = dev
        self.n = (_dev_f32(g("norm.weight"), dev), _dev_f32(g("norm.bias"), dev))
        self.wqk = _dev_bf16(torch.cat([g("q.weight")[:, :, 0, 0], g("k.weight")[:, :, 0, 0]], 0), dev)
        self.bqk = _dev_f32(torch.cat([g("q.bias"), g("k.bias")], 0), dev)
        self.wv, self.bv = _dev_bf16(g("v.weight")[:, :, 0, 0], dev), _dev_f32(g("v.bias"), dev)
        self.wo, self.bo = _dev_bf16(g("proj_out.weight")[:, :, 0, 0], dev), _dev_f32(g("proj_out.bias"), dev)

    def forward(self, x, F, H, W):
        c, pix = self.c, H * W
        assert pix % 64 == 0, "VAE mid attention expects H*W to be a multiple of 64"
        h = ops.groupnorm(x, F, pix, *self.n, 1e-6, silu=False)
        e16 = h.dtype                                                   # x may be the fp32 residual stream: every attention operand is 16 bit
        qk = ops.gemm(h, self.wqk, bias=self.bqk)                       # [F*pix, 2C]
        key = (F, pix, e16)
        vt = self._vt.get(key)
        if vt is None:
            vt = torch.zeros((F, c, pix), dtype=e16, device=self.dev)
            self._vt[key] = vt
        ops.gemm(h, self.wv, bias=self.bv, trans_out=dict(tok_per_frame=pix, tokens_ld=pix, out=vt))
        o = torch.empty((F * pix, c), dtype=e16, device=x.device)
        s = torch.empty((pix, pix), dtype=torch.float32, device=x.device)
        p = torch.empty((pix, pix), dtype=e16, device=x.device)
        for f in range(F):
            q_f, k_f = qk[f * pix:(f + 1) * pix, :c], qk[f * pix:(f + 1) * pix, c:]
            ops.gemm(q_f, k_f, out=s)                                    # scores = q k^T  (fp32)
            ops.softmax_rows(s, p, c ** -0.5)
            ops.gemm(p, vt[f], out=o[f * pix:(f + 1) * pix])             # o = P V  (W operand = V^T)
        return ops.gemm(o, self.wo, bias=self.bo, residual=x, out_f32=x.dtype == torch.float32)


class VideoDecoder:
    """temporal_ae.py:291-347 / model.py:603-748.  ``forward(z, timesteps=n)`` keeps the reference contract:
    z [n, 4, h, w] fp32 (already divided by the scale factor) -> [n, 3, 8h, 8w] fp32."""

    def __init__(self, cfg=None):
        cfg = cfg or VaeConfig()
        self.cfg = cfg
        nres = len(cfg.ch_mult)
        block_in = cfg.ch * cfg.ch_mult[-1]
        self.conv_in = _Conv("conv_in.", cfg.z_channels, block_in, x3=True)        # rim: split-3 operands under the decoder's precision plan
        self.mid_block_1 = AEVideoResBlock("mid.block_1.", block_in, block_in, stream=True)
        self.mid_attn_1 = AEAttnBlock("mid.attn_1.", block_in)
        self.mid_block_2 = AEVideoResBlock("mid.block_2.", block_in, block_in, stream=True)
        self.up = {}
        for lvl in reversed(range(nres)):
            block_out = cfg.ch * cfg.ch_mult[lvl]
            blocks = []
            for b in range(cfg.num_res_blocks + 1):
                blocks.append(AEVideoResBlock(f"up.{lvl}.block.{b}.", block_in, block_out, stream=True))
                block_in = block_out
            ups = _Conv(f"up.{lvl}.upsample.conv.", block_in, block_in, ups=1) if lvl != 0 else None
            self.up[lvl] = (blocks, ups)
        self.final_ch = block_in
        self.conv_out = _Conv("conv_out.", block_in, cfg.out_ch)
        self.prepared = False

    def _modules(self):
        yield self.conv_in
        yield self.mid_block_1
        yield self.mid_attn_1
        yield self.mid_block_2
        for lvl in self.up:
            blocks, ups = self.up[lvl]
            yield from blocks
            if ups is not None:
                yield ups
        yield self.conv_out

    def spec(self):
        s = Spec()
        for m in self._modules():
            m.spec(s)
        s.add("norm_out.weight", self.final_ch); s.add("norm_out.bias", self.final_ch)
        s.add("conv_out.time_mix_conv.weight", self.cfg.out_ch, self.cfg.out_ch, 3, 1, 1)
        s.add("conv_out.time_mix_conv.bias", self.cfg.out_ch)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        check_state_dict(self.spec(), sd)
        for m in self._modules():
            m.prepare(sd, device)
        self.no = (_dev_f32(sd["norm_out.weight"], device), _dev_f32(sd["norm_out.bias"], device))
        assert self.cfg.out_ch == 3
        self.tmw = _dev_f32(sd["conv_out.time_mix_conv.weight"][:, :, :, 0, 0], device)   # [co, ci, kt]
        self.tmb = _dev_f32(sd["conv_out.time_mix_conv.bias"], device)
        # decoder precision plan (ops.AE_EXACT_RIM, read here): norm_out + SiLU + conv_out (-> 3 channels) as the fp32 head kernel: weights [tap][c][4]
        self.head_wt = None
        if ops.AE_EXACT_RIM and self.final_ch % 32 == 0:
            wt = torch.zeros(3, 3, self.final_ch, 4, dtype=torch.float32)
            wt[..., : self.cfg.out_ch] = sd["conv_out.weight"].detach().float().permute(2, 3, 1, 0)
            self.head_wt = wt.reshape(9, self.final_ch, 4).contiguous().to(device)
            self.head_b = _dev_f32(pad_rows(sd["conv_out.bias"].detach().float(), 4), device)
        self.rim_stem = ops.AE_EXACT_RIM and self.conv_in.w3 is not None
        self.device = device
        self.prepared = True
        return self

    def forward(self, z, timesteps=None, clamp=False):
        n, _, H, W = z.shape
        assert timesteps is None or timesteps == n, "one call decodes one temporal group (streaming_svd.py:138-146)"
        F = n
        st = ops.ae_stream_on(self.mid_block_1.cout)
        if self.rim_stem:
            h = ops.nchw_to_tokens_x3(z.float().contiguous(), None, None, 32)
            h, H, W = self.conv_in.forward(h, F, H, W, split3=True, out_f32=st)
        else:
            h = ops.nchw_to_tokens(z.float().contiguous(), None, None, 32)
            h, H, W = self.conv_in.forward(h, F, H, W, out_f32=st)
        h = self.mid_block_1.forward(h, F, H, W)
        h = self.mid_attn_1.forward(h, F, H, W)
        h = self.mid_block_2.forward(h, F, H, W)
        levels = list(self.up)
        for i, lvl in enumerate(levels):
            blocks, ups = self.up[lvl]
            for b in blocks:
                h = b.forward(h, F, H, W)
            if ups is not None:        # Upsample.conv output continues the stream when the NEXT level carries it
                h, H, W = ups.forward(h, F, H, W, out_f32=ops.ae_stream_on(self.up[levels[i + 1]][0][0].cout))
        if self.head_wt is not None:
            h = ops.head_gn_silu_conv3x3(h, F, H, W, self.no[0], self.no[1], 1e-6, self.head_wt, self.head_b, self.cfg.out_ch)
        else:
            h = ops.groupnorm(h, F, H * W, *self.no, 1e-6, silu=True)
            h, _, _ = self.conv_out.forward(h, F, H, W, out_f32=True)      # [F*H*W, 4] fp32 (3 valid channels)
        return ops.ae_time_mix3(h, self.tmw, self.tmb, F, H, W, clamp)       # AE3DConv.time_mix_conv -> NCHW fp32


class AutoencodingEngineDecoder:
    """The slice of AutoencodingEngine the hot path touches: ``decode(z, timesteps=n)`` and ``.decoder``
    (code/models/svd/sgm/models/autoencoder.py:210-212, diffusion_trainer/streaming_svd.py:138-146)."""

    decode_group = None      # torch.distributed group over which StreamingSVD.decode_first_stage shards the independent frame groups

    def __init__(self, decoder):
        self.decoder = decoder

    def decode(self, z, **kwargs):
        return self.decoder.forward(z, **kwargs)

    def output_shape(self, z):
        """decoded frames of latents z [n, 4, h, w]: [n, 3, f h, f w] with f = 2^(levels - 1) (8 for the shipped 4-level decoder).  Used by
        the frame-group-sharded decode on ranks that own no group (found by tests/test_gpu_multiproc.py: a 2-level decoder is not 8x)."""
        f = 2 ** (len(self.decoder.cfg.ch_mult) - 1)
        return (z.shape[0], self.decoder.cfg.out_ch, f * z.shape[2], f * z.shape[3])


# ------------------------------------------------------------------------------------------------------------------------------
# VAE encoder (first piece): what the reference's conditioner runs on the conditioning frame to obtain the
# `concat` latents (VideoPredictionEmbedderWithEncoder -> first_stage_model.encoder, config.yaml:205-238).
class AEResBlock2D:
    """ResnetBlock with temb=None (diffusionmodules/model.py:94-151): GN(1e-6)+SiLU -> conv -> GN+SiLU -> conv (+ 1x1 shortcut)."""

    def __init__(self, prefix, cin, cout):
        self.p, self.cin, self.cout = prefix, cin, cout

    def spec(self, s):
        p, ci, co = self.p, self.cin, self.cout
        s.add(p + "norm1.weight", ci); s.add(p + "norm1.bias", ci)
        s.add(p + "conv1.weight", co, ci, 3, 3); s.add(p + "conv1.bias", co)
        s.add(p + "norm2.weight", co); s.add(p + "norm2.bias", co)
        s.add(p + "conv2.weight", co, co, 3, 3); s.add(p + "conv2.bias", co)
        if ci != co:
            s.add(p + "nin_shortcut.weight", co, ci, 1, 1); s.add(p + "nin_shortcut.bias", co)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        Fv = lambda k: _dev_f32(g(k), dev)
        self.n1, self.n2 = (Fv("norm1.weight"), Fv("norm1.bias")), (Fv("norm2.weight"), Fv("norm2.bias"))
        self.w1, self.b1 = _dev_bf16(pack_conv3x3(g("conv1.weight")), dev), Fv("conv1.bias")
        self.w2, self.b2 = _dev_bf16(pack_conv3x3(g("conv2.weight")), dev), Fv("conv2.bias")
        if self.cin != self.cout:
            self.ws, self.bs = _dev_bf16(g("nin_shortcut.weight")[:, :, 0, 0], dev), Fv("nin_shortcut.bias")

    def forward(self, x, F, H, W):
        pix = H * W
        h = ops.groupnorm(x, F, pix, *self.n1, 1e-6, silu=True)
        h = ops.gemm(h, self.w1, bias=self.b1, conv=dict(cin=self.cin, hin=H, win=W, hout=H, wout=W, frames=F))
        h = ops.groupnorm(h, F, pix, *self.n2, 1e-6, silu=True)
        skip = x if self.cin == self.cout else ops.gemm(x, self.ws, bias=self.bs)
        return ops.gemm(h, self.w2, bias=self.b2, residual=skip, conv=dict(cin=self.cout, hin=H, win=W, hout=H, wout=W, frames=F))


class _DownsampleAsym(_Conv):
    """Downsample (diffusionmodules/model.py:73-92): F.pad(x, (0,1,0,1)) then 3x3 conv, stride 2, padding 0."""

    def __init__(self, prefix, ch):
        super().__init__(prefix, ch, ch, stride=2)

    def forward(self, x, F, H, W, **kw):
        ho, wo = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
        cv = dict(cin=self.cin_pad, hin=H, win=W, hout=ho, wout=wo, stride=2, ups=0, pad_mode=1, frames=F)
        return ops.gemm(x, self.w, bias=self.b, conv=cv, **kw), ho, wo


class Encoder:
    """sgm Encoder (diffusionmodules/model.py:487-601), encoder_config of config.yaml:222-238: ``forward(x [n,3,H,W] in [-1,1])``
    -> moments [n, 2*z_channels, H/8, W/8] fp32 (mean | logvar); state_dict keys as under ``first_stage_model.encoder.``."""

    def __init__(self, cfg=None, in_channels=3, double_z=True):
        cfg = cfg or VaeConfig()
        self.cfg, self.in_channels = cfg, in_channels
        self.zc = (2 if double_z else 1) * cfg.z_channels
        self.conv_in = _Conv("conv_in.", in_channels, cfg.ch)
        self.down = []
        block_in = cfg.ch
        for lvl, mult in enumerate(cfg.ch_mult):
            blocks = []
            for b in range(cfg.num_res_blocks):
                blocks.append(AEResBlock2D(f"down.{lvl}.block.{b}.", block_in, cfg.ch * mult))
                block_in = cfg.ch * mult
            ds = _DownsampleAsym(f"down.{lvl}.downsample.conv.", block_in) if lvl != len(cfg.ch_mult) - 1 else None
            self.down.append((blocks, ds))
        self.mid_block_1 = AEResBlock2D("mid.block_1.", block_in, block_in)
        self.mid_attn_1 = AEAttnBlock("mid.attn_1.", block_in)
        self.mid_block_2 = AEResBlock2D("mid.block_2.", block_in, block_in)
        self.final_ch = block_in
        self.conv_out = _Conv("conv_out.", block_in, self.zc)
        self.prepared = False

    def _modules(self):
        yield self.conv_in
        for blocks, ds in self.down:
            yield from blocks
            if ds is not None:
                yield ds
        yield self.mid_block_1
        yield self.mid_attn_1
        yield self.mid_block_2
        yield self.conv_out

    def spec(self):
        s = Spec()
        for m in self._modules():
            m.spec(s)
        s.add("norm_out.weight", self.final_ch); s.add("norm_out.bias", self.final_ch)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        check_state_dict(self.spec(), sd)
        for m in self._modules():
            m.prepare(sd, device)
        self.no = (_dev_f32(sd["norm_out.weight"], device), _dev_f32(sd["norm_out.bias"], device))
        self.device, self.prepared = device, True
        return self

    def forward(self, x):
        n, _, H, W = x.shape
        h = ops.nchw_to_tokens(x.float().contiguous(), None, None, 32)
        h, H, W = self.conv_in.forward(h, n, H, W)
        for blocks, ds in self.down:
            for b in blocks:
                h = b.forward(h, n, H, W)
            if ds is not None:
                h, H, W = ds.forward(h, n, H, W)
        h = self.mid_block_1.forward(h, n, H, W)
        h = self.mid_attn_1.forward(h, n, H, W)
        h = self.mid_block_2.forward(h, n, H, W)
        h = ops.groupnorm(h, n, H * W, *self.no, 1e-6, silu=True)
        h, _, _ = self.conv_out.forward(h, n, H, W, out_f32=True)
        return ops.tokens_to_nchw(h, self.zc, n, H, W)

    __call__ = forward


class CondFrameEncoder:
    """AutoencoderKLModeOnly.encode (sgm/models/autoencoder.py:468-490, regulariser DiagonalGaussianRegularizer(sample=False)), the
    encoder of the conditioner's `cond_frames` embedder (config.yaml:183-214): encoder -> quant_conv (1x1, 8 -> 8) -> mode = the
    mean half.  state_dict keys ``encoder.*`` + ``quant_conv.*`` (as under ``conditioner.embedders.3.encoder.``; its decoder /
    post_quant_conv keys are not used by the conditioner and are ignored).  forward(x [n,3,H,W] in [-1,1]) -> [n, 4, H/8, W/8]."""

    def __init__(self, cfg=None):
        self.enc = Encoder(cfg)
        self.zc = self.enc.zc

    def spec(self):
        s = Spec()
        for n, sh in self.enc.spec():
            s.add("encoder." + n, *sh)
        s.add("quant_conv.weight", self.zc, self.zc, 1, 1); s.add("quant_conv.bias", self.zc)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else dict(sd)
        sd = {k: v for k, v in sd.items() if k.startswith("encoder.") or k.startswith("quant_conv.")}
        check_state_dict(self.spec(), sd)
        self.enc.load_state_dict(sd, device=device, prefix="encoder.")
        w = torch.zeros(self.zc, 32)
        w[:, : self.zc] = sd["quant_conv.weight"].detach().float()[:, :, 0, 0]
        self.qw, self.qb = _dev_bf16(w, device), _dev_f32(sd["quant_conv.bias"], device)
        self.device = device
        return self

    def forward(self, x, return_moments=False):
        e = self.enc
        n, _, H, W = x.shape
        h = ops.nchw_to_tokens(x.float().contiguous(), None, None, 32)
        h, H, W = e.conv_in.forward(h, n, H, W)
        for blocks, ds in e.down:
            for b in blocks:
                h = b.forward(h, n, H, W)
            if ds is not None:
                h, H, W = ds.forward(h, n, H, W)
        h = e.mid_block_2.forward(e.mid_attn_1.forward(e.mid_block_1.forward(h, n, H, W), n, H, W), n, H, W)
        h = ops.groupnorm(h, n, H * W, *e.no, 1e-6, silu=True)
        buf = torch.zeros((n * H * W, 32), dtype=h.dtype, device=h.device)       # moments in channels 0..7, zero padded to K = 32
        e.conv_out.forward(h, n, H, W, out=buf[:, : self.zc])
        m = ops.gemm(buf, self.qw, bias=self.qb, out_f32=True)                    # quant_conv
        if return_moments:
            return ops.tokens_to_nchw(m, self.zc, n, H, W)                         # [n, 8, h, w]: mean | logvar
        return ops.tokens_to_nchw(m, self.zc // 2, n, H, W)                        # mode of the diagonal Gaussian = mean channels

    __call__ = forward

    def sample(self, x, generator=None):
        """DiagonalGaussianDistribution.sample (diffusers vae.py / sgm distributions): mean + exp(0.5 * clamp(logvar, -30, 20)) * N(0, 1)."""
        mom = self.forward(x, return_moments=True)
        mean, logvar = mom.chunk(2, 1)
        std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
        return mean + std * torch.randn(mean.shape, generator=generator, device=mean.device)


# ------------------------------------------------------------------------------------------------------------------------------
# 2-D KL autoencoder (LDM / Stable-Diffusion VAE): the arithmetic of sgm's Encoder / Decoder (diffusionmodules/model.py:487-748),
# which diffusers' AutoencoderKL -- the VAE around the I2VGen-XL enhancer (pipeline_i2vgen_xl.py:384-406, 586-603) -- re-keys.
class Decoder2D:
    """sgm Decoder (model.py:603-748): conv_in, mid (res, attn, res), 4 up levels x (num_res_blocks + 1) res blocks + nearest-2x
    upsample conv, GN + SiLU + conv_out.  forward(z [n, 4, h, w]) -> [n, 3, 8h, 8w] fp32."""

    def __init__(self, cfg=None):
        cfg = cfg or VaeConfig()
        self.cfg = cfg
        nres = len(cfg.ch_mult)
        block_in = cfg.ch * cfg.ch_mult[-1]
        self.conv_in = _Conv("conv_in.", cfg.z_channels, block_in)
        self.mid_block_1 = AEResBlock2D("mid.block_1.", block_in, block_in)
        self.mid_attn_1 = AEAttnBlock("mid.attn_1.", block_in)
        self.mid_block_2 = AEResBlock2D("mid.block_2.", block_in, block_in)
        self.up = {}
        for lvl in reversed(range(nres)):
            block_out = cfg.ch * cfg.ch_mult[lvl]
            blocks = []
            for b in range(cfg.num_res_blocks + 1):
                blocks.append(AEResBlock2D(f"up.{lvl}.block.{b}.", block_in, block_out))
                block_in = block_out
            self.up[lvl] = (blocks, _Conv(f"up.{lvl}.upsample.conv.", block_in, block_in, ups=1) if lvl != 0 else None)
        self.final_ch = block_in
        self.conv_out = _Conv("conv_out.", block_in, cfg.out_ch)

    def _modules(self):
        yield self.conv_in; yield self.mid_block_1; yield self.mid_attn_1; yield self.mid_block_2
        for lvl in self.up:
            blocks, ups = self.up[lvl]
            yield from blocks
            if ups is not None:
                yield ups
        yield self.conv_out

    def spec(self):
        s = Spec()
        for m in self._modules():
            m.spec(s)
        s.add("norm_out.weight", self.final_ch); s.add("norm_out.bias", self.final_ch)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        check_state_dict(self.spec(), sd)
        for m in self._modules():
            m.prepare(sd, device)
        self.no = (_dev_f32(sd["norm_out.weight"], device), _dev_f32(sd["norm_out.bias"], device))
        self.device = device
        return self

    def forward(self, z, clamp=False):
        n, _, H, W = z.shape
        h = ops.nchw_to_tokens(z.float().contiguous(), None, None, 32)
        h, H, W = self.conv_in.forward(h, n, H, W)
        h = self.mid_block_2.forward(self.mid_attn_1.forward(self.mid_block_1.forward(h, n, H, W), n, H, W), n, H, W)
        for lvl in self.up:
            blocks, ups = self.up[lvl]
            for b in blocks:
                h = b.forward(h, n, H, W)
            if ups is not None:
                h, H, W = ups.forward(h, n, H, W)
        h = ops.groupnorm(h, n, H * W, *self.no, 1e-6, silu=True)
        h, _, _ = self.conv_out.forward(h, n, H, W, out_f32=True)
        return ops.tokens_to_nchw(h, self.cfg.out_ch, n, H, W)

    __call__ = forward


def diffusers_vae_to_sgm_keys(sd, n_levels=4):
    """diffusers AutoencoderKL state_dict (the `vae/` of ali-vilab/i2vgen-xl, pipeline_i2vgen_xl.py:384-406) -> the sgm / LDM key
    names of Encoder / Decoder2D (same architecture: diffusers' AutoencoderKL is the re-keyed LDM autoencoder; its attention's
    Linear to_q/to_k/to_v/to_out.0 are the 1x1 convs q/k/v/proj_out).  diffusers is not installed: key names as published in
    diffusers==0.30.2 (models/autoencoders/vae.py, unet_2d_blocks.py) -- the ARITHMETIC behind them is pinned through the sgm classes."""
    out = {}
    for k, v in sd.items():
        part, _, rest = k.partition(".")
        if part in ("quant_conv", "post_quant_conv"):
            out[k] = v
            continue
        if part not in ("encoder", "decoder"):
            continue
        r = rest
        r = r.replace("conv_norm_out.", "norm_out.").replace("conv_shortcut.", "nin_shortcut.")
        r = r.replace("mid_block.resnets.0.", "mid.block_1.").replace("mid_block.resnets.1.", "mid.block_2.")
        if r.startswith("mid_block.attentions.0."):
            r = r.replace("mid_block.attentions.0.", "mid.attn_1.").replace("group_norm.", "norm.").replace("to_out.0.", "proj_out.")
            r = r.replace("to_q.", "q.").replace("to_k.", "k.").replace("to_v.", "v.")
            if r.endswith("weight") and v.dim() == 2:
                v = v[:, :, None, None]
        if r.startswith("down_blocks."):
            _, i, kind, j, tail = r.split(".", 4)
            r = f"down.{i}.block.{j}.{tail}" if kind == "resnets" else f"down.{i}.downsample.{tail}"
        if r.startswith("up_blocks."):
            _, i, kind, j, tail = r.split(".", 4)
            lvl = n_levels - 1 - int(i)
            r = f"up.{lvl}.block.{j}.{tail}" if kind == "resnets" else f"up.{lvl}.upsample.{tail}"
        out[part + "." + r] = v
    return out


class AutoencoderKL2D:
    """The 2-D VAE around the enhancer: encode (posterior mode or sample) and decode, scaling_factor 0.18215
    (pipeline_i2vgen_xl.py:586-603 retrieve_latents(vae.encode(x)) * scaling_factor; :384-406 vae.decode(latents / scaling_factor))."""

    def __init__(self, cfg=None, scaling_factor=0.18215):
        self.enc, self.dec, self.sf = CondFrameEncoder(cfg), Decoder2D(cfg), scaling_factor

    def spec(self):
        s = Spec()
        for n, sh in self.enc.spec():
            s.add(n, *sh)
        for n, sh in self.dec.spec():
            s.add("decoder." + n, *sh)
        zc = self.dec.cfg.z_channels
        s.add("post_quant_conv.weight", zc, zc, 1, 1); s.add("post_quant_conv.bias", zc)
        return s

    def load_state_dict(self, sd, device="cuda", diffusers_keys=False):
        if diffusers_keys:
            sd = diffusers_vae_to_sgm_keys(sd, len(self.dec.cfg.ch_mult))
        check_state_dict(self.spec(), sd)
        self.enc.load_state_dict(sd, device=device)
        self.dec.load_state_dict(sd, device=device, prefix="decoder.")
        zc = self.dec.cfg.z_channels
        w = torch.zeros(zc, 32)
        w[:, :zc] = sd["post_quant_conv.weight"].detach().float()[:, :, 0, 0]
        self.pw, self.pb = _dev_bf16(w, device), _dev_f32(sd["post_quant_conv.bias"], device)
        self.device = device
        return self

    def encode_sample(self, x, generator=None):
        """retrieve_latents(vae.encode(x), generator) * scaling_factor  (pipeline_i2vgen_xl.py:488-489, 586-603)."""
        return self.enc.sample(x, generator) * self.sf

    def encode_mode(self, x):
        """x [n, 3, H, W] in [-1, 1] -> scaling_factor * mean of the posterior [n, 4, H/8, W/8]."""
        return self.enc(x) * self.sf

    def decode(self, latents):
        """latents [n, 4, h, w] (scaled) -> images [n, 3, 8h, 8w] fp32."""
        z = latents.float() * (1.0 / self.sf)
        n, zc, H, W = z.shape
        t = ops.nchw_to_tokens(z.contiguous(), None, None, 32)
        t = ops.gemm(t, self.pw, bias=self.pb, out_f32=True)                      # post_quant_conv (1x1)
        return self.dec(ops.tokens_to_nchw(t, zc, n, H, W))
