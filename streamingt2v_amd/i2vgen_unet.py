"""I2VGen-XL enhancer UNet on the HIP kernel family -- host-side mirror of the reference's
``I2VGenXLUNet`` (code/i2v_enhance/unet_i2vgen_xl.py:163-814) and of the blocks it is wired from
(unet_3d_blocks.py:274-900, transformer_2d.py:479-527, transformer_temporal.py:121-200, attention.py:414-534), with the leaf
layers of diffusers==0.30.2 (ResnetBlock2D, TemporalConvLayer, Attention/AttnProcessor2_0, GEGLU, Downsample2D, Upsample2D).

Same constructor defaults, same state_dict keys and shapes (``spec()``; checked against the vendored module by
oracle/make_golden_i2v.py), same ``forward(sample, timestep, fps, image_latents, image_embeddings, encoder_hidden_states)``.
All arithmetic runs in libsvdhip.so; torch is used for buffers and views only.  Activations are channels-last tokens
``[(b f) h w, C]`` in the 16-bit element type, so

  * the spatial transformer (self-attention over h*w tokens, cross-attention to the 77 + 64 + 4 context tokens) and the
    temporal transformer (two self-attentions over the f frames of a pixel) read the SAME buffer -- the reference's
    ``(b f) c h w <-> (b h w) f c`` permutes never happen;
  * TemporalConvLayer is four (GroupNorm over (c/32, f, h, w) + SiLU) -> 3-tap implicit GEMMs with the identity add fused;
  * everything that does not depend on the sample or the timestep (fps embedding, the 145 context tokens and their per-layer
    K / V^T projections, the processed image latents) is computed once per chunk by ``set_conditioning`` and reused by the
    29 DDIM steps -- the reference recomputes it in every forward (unet_i2vgen_xl.py:655-712).

Precision plan (round 5, ops.I2V_EXACT_RIM / ops.I2V_STREAM_F32_MIN_CH; the round-4 plan of video_model.py applied here): the tensors the
reference's residual additions run on (ResnetBlock2D output, TemporalConvLayer's identity add, the transformers' x + attn(x) / x + ff(x)
chains and block outputs, the samplers' outputs) stay fp32 BETWEEN kernels -- GEMM epilogues read / write them in fp32, norms read fp32 and
write the 16-bit operand -- and the rim (conv_in, the embedding MLPs, the image-latent projection, the 4-channel head) runs with split-3
operands / as the fp32 head kernel.  Only what a matrix core consumes is rounded to 16 bit.
"""
import torch

from . import ops
from .params import Spec, check_state_dict
from .video_model import FeedForward, RowProj, WideProj, _dev_bf16, _dev_f32, _gn_pooled, _spec_ln, pack_conv3x3, pack_tconv3, pack_x3, pad_rows


def _pad32(c):
    return (c + 31) // 32 * 32


class I2VConfig:
    """Constructor arguments of the reference's I2VGenXLUNet (unet_i2vgen_xl.py:188-211).  attn_levels[i] <=> down block i is
    a CrossAttnDownBlock3D (and up block n-1-i a CrossAttnUpBlock3D)."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 norm_num_groups=32, cross_attention_dim=1024, attention_head_dim=64, attn_levels=(True, True, True, False), sample_size=None):
        assert norm_num_groups == 32 and attention_head_dim == 64, "kernels are built for 32 groups / head dim 64"
        self.in_channels, self.out_channels = in_channels, out_channels
        # `unet.config.sample_size`: read by the pipeline for its DEFAULT height / width only (pipeline_i2vgen_xl.py:731-732: height or
        # config.sample_size * vae_scale_factor); None like the shipped ali-vilab/i2vgen-xl unet/config.json ("sample_size": null is what the
        # checkpoint carries as far as it is known offline) -- i2v_enhance_interface.py always passes height / width explicitly.
        self.sample_size = sample_size
        self.block_out_channels = tuple(block_out_channels)
        self.layers_per_block = layers_per_block
        self.cross_attention_dim = cross_attention_dim
        self.attn_levels = tuple(attn_levels)[: len(self.block_out_channels)]
        assert len(self.attn_levels) == len(self.block_out_channels)


def _conv(x, w, b, cin_pad, F, H, W, ho=None, wo=None, stride=1, ups=0, **kw):
    ho, wo = ho or H, wo or W
    return ops.gemm(ops.to_elem_rows(x), w, bias=b, conv=dict(cin=cin_pad, hin=H, win=W, hout=ho, wout=wo, stride=stride, ups=ups, frames=F), **kw)


class _Resnet:
    """diffusers ResnetBlock2D (eps 1e-5, SiLU, default time-embedding add, output_scale_factor 1)."""

    def __init__(self, p, cin, cout, temb):
        self.p, self.cin, self.cout, self.temb = p, cin, cout, temb

    def spec(self, s):
        p, ci, co = self.p, self.cin, self.cout
        _spec_ln(s, p + "norm1", ci)
        s.add(p + "conv1.weight", co, ci, 3, 3); s.add(p + "conv1.bias", co)
        s.add(p + "time_emb_proj.weight", co, self.temb); s.add(p + "time_emb_proj.bias", co)
        _spec_ln(s, p + "norm2", co)
        s.add(p + "conv2.weight", co, co, 3, 3); s.add(p + "conv2.bias", co)
        if ci != co:
            s.add(p + "conv_shortcut.weight", co, ci, 1, 1); s.add(p + "conv_shortcut.bias", co)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        self.n1 = (_dev_f32(g("norm1.weight"), dev), _dev_f32(g("norm1.bias"), dev))
        self.n2 = (_dev_f32(g("norm2.weight"), dev), _dev_f32(g("norm2.bias"), dev))
        self.w1, self.b1 = _dev_bf16(pack_conv3x3(g("conv1.weight")), dev), _dev_f32(g("conv1.bias"), dev)
        self.w2, self.b2 = _dev_bf16(pack_conv3x3(g("conv2.weight")), dev), _dev_f32(g("conv2.bias"), dev)
        self.we, self.be = _dev_bf16(g("time_emb_proj.weight"), dev), _dev_f32(g("time_emb_proj.bias"), dev)
        if self.cin != self.cout:
            self.ws, self.bs = _dev_bf16(g("conv_shortcut.weight")[:, :, 0, 0], dev), _dev_f32(g("conv_shortcut.bias"), dev)

    def forward(self, x, emb_silu, F, Fr, H, W):
        pix = H * W
        st = ops.i2v_stream_on(self.cout)       # the block output is the residual stream: fp32 between kernels when set (the input is whatever the block before wrote)
        h = ops.groupnorm(x, F, pix, *self.n1, 1e-5, silu=True)
        e = ops.gemm(emb_silu, self.we, bias=self.be, out_f32=True)                       # [B, cout]: one vector per batch element
        h = _conv(h, self.w1, self.b1, self.cin, F, H, W, rowvec=e, rows_per_vec=Fr * pix)
        h = ops.groupnorm(h, F, pix, *self.n2, 1e-5, silu=True)
        skip = x if self.cin == self.cout else ops.gemm(ops.to_elem_rows(x), self.ws, bias=self.bs, out_f32=st)
        return _conv(h, self.w2, self.b2, self.cout, F, H, W, residual=skip, out_f32=st)


class _TemporalConv:
    """diffusers TemporalConvLayer: 4 x [GroupNorm(32, eps 1e-5) over (c/32, f, h, w), SiLU, Conv3d (3,1,1)] + identity."""

    def __init__(self, p, ch):
        self.p, self.c = p, ch

    def spec(self, s):
        for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
            _spec_ln(s, f"{self.p}{name}.0", self.c)
            s.add(f"{self.p}{name}.{ci}.weight", self.c, self.c, 3, 1, 1); s.add(f"{self.p}{name}.{ci}.bias", self.c)

    def prepare(self, sd, dev):
        self.l = []
        for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
            g = lambda k: sd[f"{self.p}{name}.{k}"]
            self.l.append((_dev_f32(g("0.weight"), dev), _dev_f32(g("0.bias"), dev), _dev_bf16(pack_tconv3(g(f"{ci}.weight")), dev),
                           _dev_f32(g(f"{ci}.bias"), dev)))

    def forward(self, x, F, Fr, H, W, sp=None):
        """sp (parallel.SeqParallel): x holds this rank's FRAMES of every batch element (F = B * local frames); the layer runs in the pixel
        layout -- all Fr frames of this rank's pixel range, one all-to-all either side -- and its four norms, which pool over frames AND
        pixels, all-reduce their sums."""
        pix = H * W
        if sp is not None:
            B = F // sp.frame_counts(Fr)[sp.rank]
            x = sp.to_pixels(x, B, Fr, pix)
            pix_l, cnt = sp.pix_local(pix), float(Fr) * pix * (self.c // 32)
            F_t = B * Fr
        else:
            pix_l, F_t = pix, F
        tv = dict(cin=self.c, T=Fr, pix=pix_l)
        h = x
        for i, (nw, nb, w, b) in enumerate(self.l):
            if sp is None:
                h = ops.groupnorm(h, F, pix, nw, nb, 1e-5, frames_per_stat=Fr, silu=True)
            else:
                h = _gn_pooled(h, F_t, pix_l, nw, nb, 1e-5, Fr, cnt, sp, True)
            h = ops.gemm(h, w, bias=b, temporal=tv, residual=x if i == 3 else None, out_f32=(i == 3 and ops.i2v_stream_on(self.c)))
        return h if sp is None else sp.to_frames(h, B, Fr, pix)


def _spec_attn(s, p, c, inner, kv):
    s.add(p + "to_q.weight", inner, c); s.add(p + "to_k.weight", inner, kv); s.add(p + "to_v.weight", inner, kv)
    s.add(p + "to_out.0.weight", c, inner); s.add(p + "to_out.0.bias", c)


def _spec_block(s, b, d, kv2):
    """BasicTransformerBlock(dim d): norm1/attn1, norm2/attn2 (kv dim kv2), norm3/ff (GEGLU 4x)."""
    for n in ("norm1", "norm2", "norm3"):
        _spec_ln(s, b + n, d)
    _spec_attn(s, b + "attn1.", d, d, d)
    _spec_attn(s, b + "attn2.", d, d, kv2)
    s.add(b + "ff.net.0.proj.weight", 8 * d, d); s.add(b + "ff.net.0.proj.bias", 8 * d)
    s.add(b + "ff.net.2.weight", d, 4 * d); s.add(b + "ff.net.2.bias", d)


class _Transformer2D:
    """Transformer2DModel (continuous input, linear projections, one BasicTransformerBlock): transformer_2d.py:479-527."""

    def __init__(self, p, ch, ctx_dim):
        self.p, self.c, self.ctx, self.heads = p, ch, ctx_dim, ch // 64
        self._vt = {}
        self.kv = None

    def spec(self, s):
        p, c = self.p, self.c
        _spec_ln(s, p + "norm", c)
        s.add(p + "proj_in.weight", c, c); s.add(p + "proj_in.bias", c)
        _spec_block(s, p + "transformer_blocks.0.", c, self.ctx)
        s.add(p + "proj_out.weight", c, c); s.add(p + "proj_out.bias", c)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        W, Fv = (lambda k: _dev_bf16(g(k), dev)), (lambda k: _dev_f32(g(k), dev))
        self.dev = dev
        self.n = (Fv("norm.weight"), Fv("norm.bias"))
        self.bpi, self.bpo = Fv("proj_in.bias"), Fv("proj_out.bias")
        b = "transformer_blocks.0."
        self.ln = {n: (Fv(b + n + ".weight"), Fv(b + n + ".bias")) for n in ("norm1", "norm2", "norm3")}
        self.wqk = WideProj(torch.cat([g(b + "attn1.to_q.weight"), g(b + "attn1.to_k.weight")], 0), dev)       # round 6: the row-resident kernel at C = 320
        self.wv, self.bo = W(b + "attn1.to_v.weight"), Fv(b + "attn1.to_out.0.bias")
        self.wq2, self.wk2, self.wv2 = W(b + "attn2.to_q.weight"), W(b + "attn2.to_k.weight"), W(b + "attn2.to_v.weight")
        self.bo2 = Fv(b + "attn2.to_out.0.bias")
        self.ff = FeedForward(g, b + "ff.", dev)
        # round 6: the block's four C -> C projections with the LayerNorm behind them (ops.rowgemm320 at C = 320, else svd_gemm + svd_layernorm)
        self.p_in, self.p_o1, self.p_o2, self.p_out = (RowProj(g(k), dev) for k in ("proj_in.weight", b + "attn1.to_out.0.weight", b + "attn2.to_out.0.weight",
                                                                                    "proj_out.weight"))

    def set_context(self, ctx_tok, ctx_pad_tok, B, n_ctx, n_pad):
        """K and V^T of the cross-attention depend only on the context: once per chunk (attention.py:488-501).
        ctx_pad_tok: the same tokens with each batch element zero-padded to n_pad (multiple of 4) rows -- the transposed-output
        GEMM writes 4 tokens per store; the pad columns of V^T are W.0 = 0 and the kernel masks keys >= n_ctx anyway."""
        k = ops.gemm(ctx_tok, self.wk2)
        tok_ld = (n_ctx + 63) // 64 * 64
        vt = torch.zeros((B, self.c, tok_ld), dtype=ctx_tok.dtype, device=ctx_tok.device)
        ops.gemm(ctx_pad_tok, self.wv2, trans_out=dict(tok_per_frame=n_pad, tokens_ld=tok_ld, out=vt))
        self.kv = (k, vt, n_ctx)

    def _vt_buf(self, F, pix, dt):
        tok_ld = (pix + 63) // 64 * 64
        b = self._vt.get((F, tok_ld, dt))
        if b is None:
            b = torch.zeros((F, self.c, tok_ld), dtype=dt, device=self.dev)
            self._vt[(F, tok_ld, dt)] = b
        return b, tok_ld

    def forward(self, x, F, Fr, H, W):
        c, pix, M = self.c, H * W, F * H * W
        st = ops.i2v_stream_on(c)               # fp32 residual stream: h and the block output are fp32 between the kernels; every GEMM / attention operand is 16 bit
        e16 = ops.ELEM if x.dtype == torch.float32 else x.dtype
        h, n1 = self.p_in(ops.groupnorm(x, F, pix, *self.n, 1e-6), bias=self.bpi, ln=self.ln["norm1"], stream=st)
        qk = self.wqk(n1)
        vt, tok_ld = self._vt_buf(F, pix, e16)
        ops.gemm(n1, self.wv, trans_out=dict(tok_per_frame=pix, tokens_ld=tok_ld, out=vt))
        a = torch.empty((M, c), dtype=e16, device=x.device)
        ops.attn_spatial(qk[:, :c], qk[:, c:], vt, a, F, pix, self.heads)
        h, n2 = self.p_o1(a, bias=self.bo, residual=h, ln=self.ln["norm2"], stream=st)
        k2, vt2, n_ctx = self.kv
        q2 = ops.gemm(n2, self.wq2)
        ops.attn_cross(q2, k2, vt2, a, F, pix, n_ctx, Fr, self.heads)
        h, n3 = self.p_o2(a, bias=self.bo2, residual=h, ln=self.ln["norm3"], stream=st)
        h = self.ff(n3, residual=h)                                           # x + ff(x) is consumed by proj_out only: a GEMM operand, 16 bit
        return self.p_out(h, bias=self.bpo, residual=x, stream=st)[0]


class _TransformerTemporal:
    """TransformerTemporalModel, double_self_attention (transformer_temporal.py:71-200); inner = heads * 64."""

    def __init__(self, p, ch, heads):
        self.p, self.c, self.heads, self.d = p, ch, heads, heads * 64

    def spec(self, s):
        p, c, d = self.p, self.c, self.d
        _spec_ln(s, p + "norm", c)
        s.add(p + "proj_in.weight", d, c); s.add(p + "proj_in.bias", d)
        _spec_block(s, p + "transformer_blocks.0.", d, d)
        s.add(p + "proj_out.weight", c, d); s.add(p + "proj_out.bias", c)

    def prepare(self, sd, dev):
        g = lambda k: sd[self.p + k]
        W, Fv = (lambda k: _dev_bf16(g(k), dev)), (lambda k: _dev_f32(g(k), dev))
        self.n = (Fv("norm.weight"), Fv("norm.bias"))
        self.bpi, self.bpo = Fv("proj_in.bias"), Fv("proj_out.bias")
        b = "transformer_blocks.0."
        self.ln = {n: (Fv(b + n + ".weight"), Fv(b + n + ".bias")) for n in ("norm1", "norm2", "norm3")}
        cat3 = lambda a: WideProj(torch.cat([g(b + a + ".to_q.weight"), g(b + a + ".to_k.weight"), g(b + a + ".to_v.weight")], 0), dev)
        self.wqkv1, self.wqkv2 = cat3("attn1"), cat3("attn2")
        self.bo1, self.bo2 = Fv(b + "attn1.to_out.0.bias"), Fv(b + "attn2.to_out.0.bias")
        self.ff = FeedForward(g, b + "ff.", dev)
        self.p_in, self.p_o1, self.p_o2, self.p_out = (RowProj(g(k), dev) for k in ("proj_in.weight", b + "attn1.to_out.0.weight", b + "attn2.to_out.0.weight",
                                                                                    "proj_out.weight"))

    def forward(self, x, F, Fr, H, W, sp=None):
        """sp (parallel.SeqParallel): the frame <-> pixel split of TransformerTemporalModel (transformer_temporal.py:121-200): the whole block is
        independent per PIXEL (attention over the frames of one pixel, row-wise norms / projections / feed-forward) except its input GroupNorm,
        which pools over frames and pixels (all-reduce of the sums).  x arrives and leaves in the frame layout."""
        d, pix = self.d, H * W
        st = ops.i2v_stream_on(self.c)
        e16 = ops.ELEM if x.dtype == torch.float32 else x.dtype
        if sp is None:
            B, pix_l = F // Fr, pix
            hn = ops.groupnorm(x, F, pix, *self.n, 1e-6, frames_per_stat=Fr)
        else:
            B, pix_l = F // sp.frame_counts(Fr)[sp.rank], sp.pix_local(pix)
            x = sp.to_pixels(x, B, Fr, pix)
            hn = _gn_pooled(x, B * Fr, pix_l, *self.n, 1e-6, Fr, float(Fr) * pix * (self.c // 32), sp, False)
        M = B * Fr * pix_l
        h, n = self.p_in(hn, bias=self.bpi, ln=self.ln["norm1"], stream=st)
        a = torch.empty((M, d), dtype=e16, device=x.device)
        for nxt, wqkv, po, bo in (("norm2", self.wqkv1, self.p_o1, self.bo1), ("norm3", self.wqkv2, self.p_o2, self.bo2)):
            qkv = wqkv(n)
            ops.attn_temporal(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], a, B, Fr, Fr, pix_l, self.heads)
            h, n = po(a, bias=bo, residual=h, ln=self.ln[nxt], stream=st)      # ... and the LayerNorm the NEXT sub-block reads
        h = self.ff(n, residual=h)                                            # consumed by proj_out only: 16 bit
        out = self.p_out(h, bias=self.bpo, residual=x, stream=st)[0]
        return out if sp is None else sp.to_frames(out, B, Fr, pix)


class _Conv3:
    """plain 3x3 conv with bias (conv_in / downsamplers.0.conv / upsamplers.0.conv / conv_out / image branches)."""

    def __init__(self, p, cin, cout, stride=1, ups=0, x3=False):
        self.p, self.cin, self.cout, self.stride, self.ups = p, cin, cout, stride, ups
        self.cin_pad, self.cout_pad = _pad32(cin), (cout + 3) // 4 * 4
        self.x3, self.w3 = x3, None          # a rim convolution: split-3 operands under ops.I2V_EXACT_RIM (at prepare time)

    def spec(self, s):
        s.add(self.p + "weight", self.cout, self.cin, 3, 3); s.add(self.p + "bias", self.cout)

    def prepare(self, sd, dev):
        wp = pack_conv3x3(sd[self.p + "weight"], self.cin_pad, self.cout_pad)
        self.w = _dev_bf16(wp, dev)
        self.b = _dev_f32(pad_rows(sd[self.p + "bias"].detach().float(), self.cout_pad), dev)
        self.w3 = pack_x3(wp, 9).to(dev) if (self.x3 and ops.I2V_EXACT_RIM) else None

    def forward(self, x, F, H, W, ho=None, wo=None, split3=False, **kw):
        """x: 16-bit or fp32-stream rows [F*H*W, cin_pad]; split3=True: SPLIT-3 rows [F*H*W, 3 * cin_pad] for a rim convolution prepared
        with split-3 weights (self.w3)."""
        if self.stride == 2:
            ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        elif self.ups:
            ho, wo = ho or 2 * H, wo or 2 * W
        else:
            ho, wo = H, W
        if split3:
            assert self.w3 is not None and x.shape[1] == 3 * self.cin_pad
            return _conv(x, self.w3, self.b, 3 * self.cin_pad, F, H, W, ho, wo, self.stride, self.ups, **kw), ho, wo
        return _conv(x, self.w, self.b, self.cin_pad, F, H, W, ho, wo, self.stride, self.ups, **kw), ho, wo


class I2VGenXLUNet:
    def __init__(self, cfg=None):
        self.cfg = cfg = cfg or I2VConfig()
        boc, L, cd, ic = cfg.block_out_channels, cfg.layers_per_block, cfg.cross_attention_dim, cfg.in_channels
        c0, te = boc[0], boc[0] * 4
        self.temb_ch = te
        self.conv_in = _Conv3("conv_in.", 2 * ic, c0, x3=True)
        self.transformer_in = _TransformerTemporal("transformer_in.", c0, 8)
        self.il_proj = [_Conv3("image_latents_proj_in.0.", 4, ic * 4, x3=True), _Conv3("image_latents_proj_in.2.", ic * 4, ic * 4, x3=True),
                        _Conv3("image_latents_proj_in.4.", ic * 4, ic, x3=True)]
        self.il_ctx = [_Conv3("image_latents_context_embedding.0.", 4, ic * 8),
                       _Conv3("image_latents_context_embedding.3.", ic * 8, ic * 16, stride=2),
                       _Conv3("image_latents_context_embedding.5.", ic * 16, cd, stride=2)]
        # down: (resnet, temp_conv, attn?, temp_attn?) per layer + optional downsampler
        self.down, self.up = [], []
        ch = c0
        for i, co in enumerate(boc):
            layers = []
            for j in range(L):
                p = f"down_blocks.{i}."
                layers.append((_Resnet(f"{p}resnets.{j}.", ch if j == 0 else co, co, te), _TemporalConv(f"{p}temp_convs.{j}.", co),
                               _Transformer2D(f"{p}attentions.{j}.", co, cd) if cfg.attn_levels[i] else None,
                               _TransformerTemporal(f"{p}temp_attentions.{j}.", co, co // 64) if cfg.attn_levels[i] else None))
            ds = _Conv3(f"down_blocks.{i}.downsamplers.0.conv.", co, co, stride=2) if i != len(boc) - 1 else None
            self.down.append((layers, ds))
            ch = co
        cm = boc[-1]
        self.mid = (_Resnet("mid_block.resnets.0.", cm, cm, te), _TemporalConv("mid_block.temp_convs.0.", cm),
                    _Transformer2D("mid_block.attentions.0.", cm, cd), _TransformerTemporal("mid_block.temp_attentions.0.", cm, cm // 64),
                    _Resnet("mid_block.resnets.1.", cm, cm, te), _TemporalConv("mid_block.temp_convs.1.", cm))
        rev, rattn = list(reversed(boc)), list(reversed(cfg.attn_levels))
        prev = rev[0]
        for i, co in enumerate(rev):
            cin_skip = rev[min(i + 1, len(boc) - 1)]
            layers = []
            for j in range(L + 1):
                p = f"up_blocks.{i}."
                skip_c = cin_skip if j == L else co
                rin = prev if j == 0 else co
                layers.append((_Resnet(f"{p}resnets.{j}.", rin + skip_c, co, te), _TemporalConv(f"{p}temp_convs.{j}.", co),
                               _Transformer2D(f"{p}attentions.{j}.", co, cd) if rattn[i] else None,
                               _TransformerTemporal(f"{p}temp_attentions.{j}.", co, co // 64) if rattn[i] else None))
            us = _Conv3(f"up_blocks.{i}.upsamplers.0.conv.", co, co, ups=1) if i != len(boc) - 1 else None
            self.up.append((layers, us))
            prev = co
        self.conv_out = _Conv3("conv_out.", c0, cfg.out_channels)
        self._const = None

    # -------------------------------------------------------------------------------------------- reference API shims
    @property
    def config(self):
        """`unet.config.in_channels` / `.cross_attention_dim` / `.sample_size` as read by the pipeline (pipeline_i2vgen_xl.py:731-732,819)."""
        return self.cfg

    def enable_forward_chunking(self, dim=0, num_chunks=1, **_ignored):
        """Memory-only knob of the reference, same signature (unet_i2vgen_xl.py:439: `(self, dim=0, num_chunks=1)`; inference_i2v.py:153 calls it with
        `dim=0, num_chunks=4` under use_memopt); results do not depend on it: no-op here."""
        return None

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    # -------------------------------------------------------------------------------------------- parameters
    def _modules(self):
        yield self.conv_in; yield self.transformer_in
        yield from self.il_proj; yield from self.il_ctx
        for layers, s in self.down + self.up:
            for t in layers:
                yield from (m for m in t if m is not None)
            if s is not None:
                yield s
        yield from self.mid
        yield self.conv_out

    def spec(self):
        s = Spec()
        cfg, te, c0, cd, ic = self.cfg, self.temb_ch, self.cfg.block_out_channels[0], self.cfg.cross_attention_dim, self.cfg.in_channels
        for m in self._modules():
            m.spec(s)
        e = "image_latents_temporal_encoder."
        _spec_ln(s, e + "norm1", ic)
        _spec_attn(s, e + "attn1.", ic, 2 * ic, ic)
        s.add(e + "ff.net.0.proj.weight", 4 * ic, ic); s.add(e + "ff.net.0.proj.bias", 4 * ic)
        s.add(e + "ff.net.2.weight", ic, 4 * ic); s.add(e + "ff.net.2.bias", ic)
        for n, (a, b) in (("time_embedding.linear_1", (te, c0)), ("time_embedding.linear_2", (te, te)), ("context_embedding.0", (te, cd)),
                          ("context_embedding.2", (cd * ic, te)), ("fps_embedding.0", (te, c0)), ("fps_embedding.2", (te, te))):
            s.add(n + ".weight", a, b); s.add(n + ".bias", a)
        _spec_ln(s, "conv_norm_out", c0)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        check_state_dict(self.spec(), sd, prefix)
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
        self.dev = device
        # `unet.dtype` / `unet.device` as the pipeline reads them (pipeline_i2vgen_xl.py:265 casts the prompt embeddings to unet.dtype): the dtype of
        # the weights this mirror was loaded from, so that the swap does not change what the pipeline hands the UNet (found by executing the swap,
        # tests/test_dropin_enhancer_reference.py)
        self.dtype = next(iter(sd.values())).dtype
        self.device = torch.device(device)
        for m in self._modules():
            m.prepare(sd, device)
        W, Fv = (lambda k: _dev_bf16(sd[k], device)), (lambda k: _dev_f32(sd[k], device))
        self.lin = {n: (W(n + ".weight"), Fv(n + ".bias")) for n in ("time_embedding.linear_1", "time_embedding.linear_2", "context_embedding.0",
                                                                    "context_embedding.2", "fps_embedding.0", "fps_embedding.2")}
        self.norm_out = (Fv("conv_norm_out.weight"), Fv("conv_norm_out.bias"))
        # precision plan (ops.I2V_EXACT_RIM): the time / fps embedding MLPs with split-3 operands (a handful of rows), the head (conv_norm_out + SiLU +
        # conv_out to 4 channels) as the fp32 kernel svd_head_gn_silu_conv3x3 with weights [tap (ky, kx)][c][4]
        self.lin3, self.head_wt = {}, None
        c0, oc = self.cfg.block_out_channels[0], self.cfg.out_channels
        if ops.I2V_EXACT_RIM:
            for n in ("time_embedding.linear_1", "time_embedding.linear_2", "fps_embedding.0", "fps_embedding.2"):
                self.lin3[n] = pack_x3(sd[n + ".weight"], 1).to(device)
            if oc <= 4 and c0 % 32 == 0:
                wt = torch.zeros(3, 3, c0, 4, dtype=torch.float32)
                wt[..., :oc] = sd["conv_out.weight"].detach().float().permute(2, 3, 1, 0)
                self.head_wt = wt.reshape(9, c0, 4).contiguous().to(device)
                self.head_b = _dev_f32(pad_rows(sd["conv_out.bias"].detach().float(), 4), device)
        e = "image_latents_temporal_encoder."
        flat = [sd[e + k].detach().float().reshape(-1) for k in ("norm1.weight", "norm1.bias", "attn1.to_q.weight", "attn1.to_k.weight",
                "attn1.to_v.weight", "attn1.to_out.0.weight", "attn1.to_out.0.bias", "ff.net.0.proj.weight", "ff.net.0.proj.bias",
                "ff.net.2.weight", "ff.net.2.bias")]
        self.enc_params = torch.cat(flat).to(device).contiguous()
        assert self.enc_params.numel() == 288, "image temporal encoder kernel is built for in_channels = 4"
        return self

    # -------------------------------------------------------------------------------------------- per-chunk constants
    def set_conditioning(self, fps, image_latents, image_embeddings, encoder_hidden_states):
        """Everything of forward() that depends neither on the sample nor on the timestep (unet_i2vgen_xl.py:655-712)."""
        B, C, Fr, H, W = image_latents.shape
        dev, cd, c0 = self.dev, self.cfg.cross_attention_dim, self.cfg.block_out_channels[0]
        lin = lambda n, x, **kw: ops.gemm(x, self.lin[n][0], bias=self.lin[n][1], **kw)
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        fps_emb = self._embed_mlp("fps_embedding.0", "fps_embedding.2", f32(fps), c0)                                         # [B, te] fp32
        # 64 context tokens from the first frame's image latents
        first = f32(image_latents[:, :, 0])
        x = ops.nchw_to_tokens(first, None, None, 32)
        x, _, _ = self.il_ctx[0].forward(x, B, H, W, silu=True)
        x = ops.adaptive_avgpool(x, B, H, W, 32, 32)
        x, h2, w2 = self.il_ctx[1].forward(x, B, 32, 32, silu=True)
        x, h3, w3 = self.il_ctx[2].forward(x, B, h2, w2)                                   # [B*64, cd]
        img = lin("context_embedding.2", lin("context_embedding.0", ops.to_elem(f32(image_embeddings).reshape(B, cd)), silu=True))   # [B, 4*cd]
        text = ops.to_elem(f32(encoder_hidden_states))
        ctx = torch.cat([text, x.view(B, h3 * w3, cd), img.view(B, -1, cd)], 1).contiguous()                                  # [B, 145, cd]
        n_ctx = ctx.shape[1]
        n_pad = (n_ctx + 3) // 4 * 4
        ctx_pad = torch.zeros((B, n_pad, cd), dtype=ctx.dtype, device=dev)
        ctx_pad[:, :n_ctx] = ctx
        kv = []
        for m in self._modules():
            if isinstance(m, _Transformer2D):
                m.set_context(ctx.view(B * n_ctx, cd), ctx_pad.view(B * n_pad, cd), B, n_ctx, n_pad)
                kv.append((m, m.kv))
        # processed image latents (concatenated to every sample)
        il = f32(image_latents.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W))
        if self.il_proj[0].w3 is not None:      # precision plan: the projection with split-3 operands, fp32 between its three convolutions
            t = ops.nchw_to_tokens_x3(il, None, None, 32)
            M = t.shape[0]
            for i, (cv, act) in enumerate(zip(self.il_proj, (True, True, False))):
                buf = torch.zeros((M, 32), dtype=torch.float32, device=dev)
                _conv(t, cv.w3, cv.b, 3 * cv.cin_pad, B * Fr, H, W, silu=act, out=buf[:, : cv.cout_pad])
                t = ops.rows_split3(buf) if i < 2 else buf
        else:
            t = ops.nchw_to_tokens(il, None, None, 32)
            M = t.shape[0]
            for cv, act in zip(self.il_proj, (True, True, False)):
                buf = torch.zeros((M, 32), dtype=t.dtype, device=dev)
                _conv(t, cv.w, cv.b, cv.cin_pad, B * Fr, H, W, silu=act, out=buf[:, : cv.cout_pad])
                t = buf
        il_out = ops.i2v_image_temporal_encoder(t, self.enc_params, B, Fr, H, W)             # fp32 [(b f), 4, H, W]
        self._const = dict(fps_emb=fps_emb, il=il_out, B=B, Fr=Fr, H=H, W=W, kv=kv)
        return self._const

    def _embed_mlp(self, n1, n2, values, dim, rowvec=None):
        """Timesteps(dim) -> Linear -> SiLU -> Linear (+ rowvec) -> fp32 rows: time_embedding / fps_embedding (unet_i2vgen_xl.py:655-668);
        split-3 operands under the precision plan."""
        (w1, b1), (w2, b2) = self.lin[n1], self.lin[n2]
        if n1 in self.lin3:
            h = ops.gemm(ops.rows_split3(ops.timestep_embedding(values, dim, f32=True)), self.lin3[n1], bias=b1, silu=True, out_f32=True)
            return ops.gemm(ops.rows_split3(h), self.lin3[n2], bias=b2, rowvec=rowvec, rows_per_vec=1 if rowvec is not None else 0, out_f32=True)
        h = ops.gemm(ops.timestep_embedding(values, dim), w1, bias=b1, silu=True)
        return ops.gemm(h, w2, bias=b2, rowvec=rowvec, rows_per_vec=1 if rowvec is not None else 0, out_f32=True)

    def use_conditioning(self, const):
        """Re-install the constants of an earlier set_conditioning call (one per blending window; they do not change over
        the DDIM steps)."""
        self._const = const
        for m, kv in const["kv"]:
            m.kv = kv
        return self

    # -------------------------------------------------------------------------------------------- forward
    sp = None      # parallel.SeqParallel: forward_frames runs frame <-> pixel sequence-parallel over its group (I2VEnhancer.denoise(plan=...))

    def forward_frames(self, sample_frames, timestep):
        """sample_frames fp32 [(b f), 4, H, W] (contiguous) -> noise prediction fp32 [(b f), 4, H, W].

        With `self.sp` (round 5; SURVEY 8e, the frame <-> pixel split of the enhancer): every rank receives ALL frames (a 38-frame window of
        4 x 90 x 160 latents is 8.8 MB), keeps its contiguous share of the frames of each batch element for the per-frame operators (ResnetBlock2D,
        Transformer2DModel with its N = 14 400 spatial attention, the samplers, the head), changes to the pixel layout with one all-to-all either side
        of every TemporalConvLayer / TransformerTemporalModel, and all-gathers the 4-channel prediction at the end: every rank returns all frames."""
        k = self._const
        B, Fr, H, W = k["B"], k["Fr"], k["H"], k["W"]
        sp = self.sp
        c0 = self.cfg.block_out_channels[0]
        tt = torch.full((B,), float(timestep), dtype=torch.float32, device=self.dev)
        emb = self._embed_mlp("time_embedding.linear_1", "time_embedding.linear_2", tt, c0, rowvec=k["fps_emb"])
        emb_silu = ops.to_elem(emb, silu=True)
        il = k["il"]
        if sp is None:
            Fs = Fr                                   # frames of one batch element among this rank's rows
        else:
            Fs = sp.frame_counts(Fr)[sp.rank]
            assert Fs > 0, f"sequence-parallel degree {sp.size} exceeds the {Fr} frames of the window"
            sample_frames = sp.take_frames(sample_frames, B, Fr)
            held = k.get("il_sp")
            if held is None or held[0] is not sp:
                k["il_sp"] = held = (sp, sp.take_frames(il, B, Fr))
            il = held[1]
        F = B * Fs

        st0 = ops.i2v_stream_on(c0)
        if self.conv_in.w3 is not None:         # precision plan: the 8-channel stem with split-3 operands
            x, _, _ = self.conv_in.forward(ops.nchw_to_tokens_x3(sample_frames, il, None, 32), F, H, W, split3=True, out_f32=st0)
        else:
            x, _, _ = self.conv_in.forward(ops.nchw_to_tokens(sample_frames, il, None, 32), F, H, W, out_f32=st0)
        x = self.transformer_in.forward(x, F, Fr, H, W, sp=sp)
        skips = [(x, H, W)]
        for layers, ds in self.down:
            for rn, tc, at, ta in layers:
                x = tc.forward(rn.forward(x, emb_silu, F, Fs, H, W), F, Fr, H, W, sp=sp)
                if at is not None:
                    x = ta.forward(at.forward(x, F, Fs, H, W), F, Fr, H, W, sp=sp)
                skips.append((x, H, W))
            if ds is not None:
                x, H, W = ds.forward(x, F, H, W, out_f32=ops.i2v_stream_on(ds.cout))      # the samplers' outputs start the next level's stream
                skips.append((x, H, W))
        r0, t0, at, ta, r1, t1 = self.mid
        x = t0.forward(r0.forward(x, emb_silu, F, Fs, H, W), F, Fr, H, W, sp=sp)
        x = ta.forward(at.forward(x, F, Fs, H, W), F, Fr, H, W, sp=sp)
        x = t1.forward(r1.forward(x, emb_silu, F, Fs, H, W), F, Fr, H, W, sp=sp)
        for layers, us in self.up:
            for rn, tc, at, ta in layers:
                s, sh, sw = skips.pop()
                assert (sh, sw) == (H, W)
                x = ops.concat_channels(x, s)
                x = tc.forward(rn.forward(x, emb_silu, F, Fs, H, W), F, Fr, H, W, sp=sp)
                if at is not None:
                    x = ta.forward(at.forward(x, F, Fs, H, W), F, Fr, H, W, sp=sp)
            if us is not None:
                # upsample_size = size of the next skip tensor (unet_i2vgen_xl.py:771-772): odd sizes are one short of 2x
                _, th, tw = skips[-1]
                x, H, W = us.forward(x, F, H, W, ho=th, wo=tw, out_f32=ops.i2v_stream_on(us.cout))
        if self.head_wt is not None:            # conv_norm_out + SiLU + conv_out as one fp32 kernel (precision plan)
            x = ops.head_gn_silu_conv3x3(x, F, H, W, *self.norm_out, 1e-5, self.head_wt, self.head_b, self.cfg.out_channels)
        else:
            x = ops.groupnorm(x, F, H * W, *self.norm_out, 1e-5, silu=True)
            x, _, _ = self.conv_out.forward(x, F, H, W, out_f32=True)
        if sp is not None:
            x = sp.gather_frames(x.contiguous(), B, Fr, H * W)
            F = B * Fr
        return ops.tokens_to_nchw(x, self.cfg.out_channels, F, H, W)

    def forward(self, sample, timestep, fps=None, image_latents=None, image_embeddings=None, encoder_hidden_states=None, **_ignored):
        """Reference signature (unet_i2vgen_xl.py:573-585): sample [B,4,F,h,w] -> (prediction [B,4,F,h,w],)."""
        if image_latents is not None:
            self.set_conditioning(fps, image_latents, image_embeddings, encoder_hidden_states)
        B, C, Fr, H, W = sample.shape
        fr = sample.to(device=self.dev, dtype=torch.float32).permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W).contiguous()
        out = self.forward_frames(fr, float(timestep))
        return (out.view(B, Fr, C, H, W).permute(0, 2, 1, 3, 4),)

    __call__ = forward
