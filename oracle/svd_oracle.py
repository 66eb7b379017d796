"""CPU oracle for the StreamingSVD denoising hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch fp32 restatement ("port") of the reference algorithm, written functionally over a state_dict with
the reference's key names.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (streamingt2v_amd/) never does and fails loudly without its HIP extension.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the unmodified reference modules from /root/reference
(imported with the stubs of oracle/ref_bootstrap.py) on seeded inputs/weights and checks this restatement against
them (max abs err <= 2e-4 in fp32), then commits small golden vectors under tests/golden/ which the CPU test-suite
re-checks.  The one un-vendored dependency on the path, diffusers==0.30.2 ``Attention`` (CAM merger), is restated
from its published behaviour in both places (see ref_bootstrap.py) -- that single op is "parity unpinned".

Every function cites the reference file:line it follows (paths relative to /root/reference/code).
Layout here is the reference's own: NCHW / (b t) c h w, fp32.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# Execution hooks (round 6).  The restatement is device-agnostic: every tensor it creates lives on its inputs' device, so the same functions
# run on CPU (tests, golden checks) and -- for oracle/make_golden_fullsize_gpu.py only -- on a GPU with STOCK PyTorch ops.  That generator swaps
# these three entry points for forms that do not depend on MIOpen's / SDPA's backend choice (im2col + matmul convolutions, explicit softmax
# attention in frame batches); on CPU they are the library calls the reference makes.
conv2d = F.conv2d
conv3d = F.conv3d
sdpa = F.scaled_dot_product_attention


# ---- small pieces ---------------------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """models/svd/sgm/modules/diffusionmodules/util.py:207-231  ([cos | sin])."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _gn(sd, p, x, eps, cast_back=False):
    """cast_back: GroupNorm32 (util.py:274-276: ``super().forward(x.float()).type(x.dtype)``) -- a no-op in fp32, the reference's rounding point
    under its shipped fp16 autocast; plain nn.GroupNorm (attention.py:132-135, conditioning.py:33-34, model.py:52-55) has none."""
    y = F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)
    return y.type(x.dtype) if cast_back else y


def _lin(sd, p, x, bias=True):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _mha(q, k, v, heads):
    """attention.py:324-347 (q,k,v [B,N,C] -> SDPA per head, scale d^-0.5)."""
    B, N, C = q.shape
    d = C // heads
    q, k, v = (t.view(B, t.shape[1], heads, d).transpose(1, 2) for t in (q, k, v))
    o = sdpa(q, k, v)
    return o.transpose(1, 2).reshape(B, N, C)


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, attention.py:288-351 (to_q/k/v without bias, to_out.0 with bias)."""
    ctx = x if context is None else context
    q = _lin(sd, p + "to_q", x, False)
    k = _lin(sd, p + "to_k", ctx, False)
    v = _lin(sd, p + "to_v", ctx, False)
    return _lin(sd, p + "to_out.0", _mha(q, k, v, heads))


def feed_forward(sd, p, x):
    """FeedForward with GEGLU, attention.py:94-120 (x, gate = proj.chunk(2); x * gelu_erf(gate))."""
    h = _lin(sd, p + "net.0.proj", x)
    a, g = h.chunk(2, dim=-1)
    return _lin(sd, p + "net.2", a * F.gelu(g))


def alpha_of(sd, name):
    """AlphaBlender 'learned_with_images' with image_only_indicator == 0: sigmoid(mix_factor). util.py:341-357."""
    return torch.sigmoid(sd[name])


def _blend(a, x_spatial, x_temporal):
    """AlphaBlender.forward, util.py:366-369: alpha.to(x_spatial.dtype) * x_spatial + (1 - alpha).to(x_spatial.dtype) * x_temporal."""
    return a.to(x_spatial.dtype) * x_spatial + (1.0 - a).to(x_spatial.dtype) * x_temporal


# ---- ResBlocks ------------------------------------------------------------------------------------------------
def res_block(sd, p, x, emb, dims=2, exchange_temb_dims=False, eps=1e-5):
    """ResBlock._forward, openaimodel.py:328-354 (no up/down, no scale-shift norm)."""
    conv = conv2d if dims == 2 else conv3d
    pad = 1 if dims == 2 else (1, 0, 0)
    h = conv(F.silu(_gn(sd, p + "in_layers.0", x, eps, True)), sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=pad)
    if emb is not None:
        e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"]).type(h.dtype)      # openaimodel.py:341
        while e.dim() < h.dim():
            e = e[..., None]
        if exchange_temb_dims:
            e = e.transpose(1, 2)            # "b t c ... -> b c t ..."
        h = h + e
    h = conv(F.silu(_gn(sd, p + "out_layers.0", h, eps, True)), sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=pad)
    if (p + "skip_connection.weight") in sd:
        x = conv(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h


def video_res_block(sd, p, x, emb, T):
    """VideoResBlock.forward, models/diffusion/video_model.py:66-85."""
    x = res_block(sd, p, x, emb)
    BT, C, H, W = x.shape
    x5 = x.view(BT // T, T, C, H, W).transpose(1, 2)                       # (b t) c h w -> b c t h w
    xt = res_block(sd, p + "time_stack.", x5, emb.view(BT // T, T, -1), dims=3, exchange_temb_dims=True)
    a = alpha_of(sd, p + "time_mixer.mix_factor")
    out = _blend(a, x5, xt)                                                  # alpha * spatial + (1-alpha) * temporal
    return out.transpose(1, 2).reshape(BT, C, H, W)


# ---- transformers ---------------------------------------------------------------------------------------------
def apm_context(sd, p, context):
    """BasicTransformerBlockWithAPM.forward, attention.py:613-619: a multi-token context [b, 17, 1024] becomes ONE effective token
    context[:, :1] + LayerNorm(Conv1d(17 -> 1, k 3, "same")(context)) * silu(apm_alpha)."""
    mixed = F.conv1d(context, sd[p + "apm_conv.weight"], sd[p + "apm_conv.bias"], padding=1)
    mixed = F.layer_norm(mixed, (context.shape[-1],), sd[p + "apm_ln.weight"], sd[p + "apm_ln.bias"], 1e-5)
    return context[:, :1] + mixed * F.silu(sd[p + "apm_alpha"])


def basic_transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:567-593 (+ the APM front of BasicTransformerBlockWithAPM when its parameters exist and
    the context carries more than one token)."""
    if context is not None and context.shape[1] > 1 and (p + "apm_conv.weight") in sd:
        context = apm_context(sd, p, context)
    x = cross_attention(sd, p + "attn1.", _ln(sd, p + "norm1", x), None, heads) + x
    x = cross_attention(sd, p + "attn2.", _ln(sd, p + "norm2", x), context, heads) + x
    return feed_forward(sd, p + "ff.", _ln(sd, p + "norm3", x)) + x


def video_transformer_block(sd, p, x, context, T, heads):
    """VideoTransformerBlock._forward, video_attention.py:125-168 (ff_in, is_res, cross-attn to time context)."""
    BT, S, C = x.shape
    b = BT // T
    x = x.view(b, T, S, C).transpose(1, 2).reshape(b * S, T, C)             # (b t) s c -> (b s) t c
    x = feed_forward(sd, p + "ff_in.", _ln(sd, p + "norm_in", x)) + x
    x = cross_attention(sd, p + "attn1.", _ln(sd, p + "norm1", x), None, heads) + x
    x = cross_attention(sd, p + "attn2.", _ln(sd, p + "norm2", x), context, heads) + x
    x = feed_forward(sd, p + "ff.", _ln(sd, p + "norm3", x)) + x
    return x.view(b, S, T, C).transpose(1, 2).reshape(BT, S, C)


def spatial_video_transformer(sd, p, x, context, T):
    """SpatialVideoTransformer.forward, video_attention.py:260-333 (use_linear, use_spatial_context, depth 1)."""
    BT, C, H, W = x.shape
    heads = C // 64
    x_in = x
    time_context = context[::T].repeat_interleave(H * W, dim=0)             # video_attention.py:281-285
    h = _gn(sd, p + "norm", x, 1e-6).flatten(2).transpose(1, 2)             # b c h w -> b (h w) c
    h = _lin(sd, p + "proj_in", h)
    t_emb = timestep_embedding(torch.arange(T, device=x.device).repeat(BT // T), C)
    emb = F.linear(F.silu(F.linear(t_emb, sd[p + "time_pos_embed.0.weight"], sd[p + "time_pos_embed.0.bias"])),
                   sd[p + "time_pos_embed.2.weight"], sd[p + "time_pos_embed.2.bias"])[:, None, :]
    h = basic_transformer_block(sd, p + "transformer_blocks.0.", h, context, heads)
    h_mix = video_transformer_block(sd, p + "time_stack.0.", h + emb, time_context, T, heads)
    a = alpha_of(sd, p + "time_mixer.mix_factor")
    h = _blend(a, h, h_mix)
    h = _lin(sd, p + "proj_out", h)
    return h.transpose(1, 2).reshape(BT, C, H, W) + x_in


def conditional_model(sd, p, sample, conditioning, T, Tc):
    """CAM merger: ConditionalModel.forward + CrossAttention.forward, models/cam/conditioning.py:117-146, 39-81.
    Inner attention = diffusers 0.30.2 Attention (heads C/64, no qkv bias, out bias)."""
    p = p + "temporal_transformer."
    BT, C, H, W = sample.shape
    B = BT // T
    heads = C // 64
    hs = sample.view(B, T, C, H, W).transpose(1, 2)                           # B C F H W
    hs = _gn(sd, p + "norm", hs, 1e-6)
    hs = hs.permute(0, 3, 4, 2, 1).reshape(B * H * W, T, C)                   # (B H W) F C
    hs = _lin(sd, p + "proj_in", hs)
    cond = conditioning.view(B, Tc, C, H, W).permute(0, 3, 4, 1, 2).reshape(B * H * W, Tc, C)
    q = _lin(sd, p + "attention.to_q", hs, False)
    k = _lin(sd, p + "attention.to_k", cond, False)
    v = _lin(sd, p + "attention.to_v", cond, False)
    a = _lin(sd, p + "attention.to_out.0", _mha(q, k, v, heads))
    r = _lin(sd, p + "proj_out", a)                                           # (B H W) F C
    r = r.view(B, H, W, T, C).permute(0, 3, 4, 1, 2).reshape(BT, C, H, W)
    return sample + r                                                         # dropout is identity in eval


# ---- networks ---------------------------------------------------------------------------------------------------
class Cfg:
    """Subset of config.yaml:69-115 the forward depends on."""

    def __init__(self, model_channels=320, num_res_blocks=2, attention_resolutions=(4, 2, 1), channel_mult=(1, 2, 4, 4),
                 cond_embed_channels=(32, 96, 256, 512)):
        self.mc, self.nrb = model_channels, num_res_blocks
        self.attn_res, self.mult = tuple(attention_resolutions), tuple(channel_mult)
        self.cond_embed_channels = tuple(cond_embed_channels)


def _emb(sd, cfg, timesteps, y):
    e = timestep_embedding(timesteps, cfg.mc)
    e = F.linear(F.silu(F.linear(e, sd["time_embed.0.weight"], sd["time_embed.0.bias"])), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    l = F.linear(F.silu(F.linear(y, sd["label_emb.0.0.weight"], sd["label_emb.0.0.bias"])), sd["label_emb.0.2.weight"], sd["label_emb.0.2.bias"])
    return e + l


def _encoder(sd, cfg, h, emb, context, T, after_stem=None):
    """input_blocks loop shared by VideoUNet.forward (video_model.py:569-579) and ControlNet.forward
    (controlnet.py:524-538)."""
    hs = []
    h = conv2d(h, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    if after_stem is not None:
        h = after_stem(h)
    hs.append(h)
    idx, ds = 1, 1
    for level, _ in enumerate(cfg.mult):
        for _ in range(cfg.nrb):
            h = video_res_block(sd, f"input_blocks.{idx}.0.", h, emb, T)
            if ds in cfg.attn_res:
                h = spatial_video_transformer(sd, f"input_blocks.{idx}.1.", h, context, T)
            hs.append(h)
            idx += 1
        if level != len(cfg.mult) - 1:
            h = conv2d(h, sd[f"input_blocks.{idx}.0.op.weight"], sd[f"input_blocks.{idx}.0.op.bias"], stride=2, padding=1)
            hs.append(h)
            idx += 1
            ds *= 2
    return hs, h, ds


def _middle(sd, h, emb, context, T):
    h = video_res_block(sd, "middle_block.0.", h, emb, T)
    h = spatial_video_transformer(sd, "middle_block.1.", h, context, T)
    return video_res_block(sd, "middle_block.2.", h, emb, T)


def video_unet(sd, cfg, x, timesteps, context, y, T, hs_control_input=None, hs_control_mid=None, Tc=None):
    """VideoUNet.forward, models/diffusion/video_model.py:540-618."""
    emb = _emb(sd, cfg, timesteps, y)
    hs, h, ds = _encoder(sd, cfg, x, emb, context, T)
    if hs_control_input is not None:
        hs = [conditional_model(sd, f"cross_attention_merger_input_blocks.{i}.", a, c, T, Tc)
              for i, (a, c) in enumerate(zip(hs, hs_control_input))]
    h = _middle(sd, h, emb, context, T)                      # consumes the UN-merged h (video_model.py:593)
    if hs_control_mid is not None:
        h = conditional_model(sd, "cross_attention_merger_mid_block.", h, hs_control_mid, T, Tc)
    idx = 0
    for level in reversed(range(len(cfg.mult))):
        for i in range(cfg.nrb + 1):
            h = torch.cat([h, hs.pop()], dim=1)
            h = video_res_block(sd, f"output_blocks.{idx}.0.", h, emb, T)
            n = 1
            if ds in cfg.attn_res:
                h = spatial_video_transformer(sd, f"output_blocks.{idx}.1.", h, context, T)
                n = 2
            if level and i == cfg.nrb:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = conv2d(h, sd[f"output_blocks.{idx}.{n}.conv.weight"], sd[f"output_blocks.{idx}.{n}.conv.bias"], padding=1)
                ds //= 2
            idx += 1
    h = h.type(x.dtype)                                                       # video_model.py:617
    h = F.silu(_gn(sd, "out.0", h, 1e-5, True))
    return conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def controlnet_cond_embedding(sd, cfg, cond):
    """ControlNetConditioningEmbedding.forward, models/control/controlnet.py:104-121 (LayerNorm over C per pixel)."""
    p = "controlnet_cond_embedding."
    e = F.silu(conv2d(cond, sd[p + "conv_in.weight"], sd[p + "conv_in.bias"], padding=1))
    nblk = 2 * (len(cfg.cond_embed_channels) - 1)
    for i in range(nblk):
        e = conv2d(e, sd[p + f"blocks.{i}.weight"], sd[p + f"blocks.{i}.bias"], padding=1, stride=2 if i % 2 else 1)
        e = F.layer_norm(e.permute(0, 2, 3, 1), (e.shape[1],), sd[p + f"norms.{i}.weight"], sd[p + f"norms.{i}.bias"], 1e-5)
        e = F.silu(e.permute(0, 3, 1, 2))
    return conv2d(e, sd[p + "conv_out.weight"], sd[p + "conv_out.bias"], padding=1)


def controlnet(sd, cfg, x, timesteps, controlnet_cond, context, y, T):
    """ControlNet.forward, models/control/controlnet.py:496-554 (Merger 'addition' after input_blocks[0])."""
    emb = _emb(sd, cfg, timesteps, y)
    cond = controlnet_cond_embedding(sd, cfg, controlnet_cond)
    hs, h, _ = _encoder(sd, cfg, x, emb, context, T, after_stem=lambda t: t + cond)
    return hs, _middle(sd, h, emb, context, T)


def streaming_wrapper(sd_unet, sd_cn, cfg, x, t, c, batch_size, num_video_frames, num_frame_conditioning, ctrl_frames,
                      use_controlnet=True):
    """StreamingWrapper.forward, models/diffusion/wrappers.py:23-78."""
    T, Tc = num_video_frames, num_frame_conditioning

    def reduce(v):
        return v.view(batch_size, T, *v.shape[1:])[:, :Tc].reshape(batch_size * Tc, *v.shape[1:])

    x = torch.cat((x, c["concat"]), dim=1)
    context, y = c["crossattn"], c["vector"]
    hs_c = mid_c = None
    if use_controlnet:
        cond = ctrl_frames.repeat(2, *([1] * (ctrl_frames.dim() - 1))).flatten(0, 1)       # (2 B) F ... -> (B F) ...
        hs_c, mid_c = controlnet(sd_cn, cfg, reduce(x), reduce(t), cond, reduce(context[:, :1]), reduce(y), Tc)
    return video_unet(sd_unet, cfg, x, t, context, y, T, hs_c, mid_c, Tc)


# ---- temporal VAE decoder ---------------------------------------------------------------------------------------
class VaeCfg:
    """decoder_config of config.yaml:241-258."""

    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, out_ch=3):
        self.ch, self.ch_mult, self.nrb, self.z, self.out_ch = ch, tuple(ch_mult), num_res_blocks, z_channels, out_ch


def _ae_resnet(sd, p, x):
    """ResnetBlock.forward with temb=None, diffusionmodules/model.py:131-151."""
    h = conv2d(F.silu(_gn(sd, p + "norm1", x, 1e-6)), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = conv2d(F.silu(_gn(sd, p + "norm2", h, 1e-6)), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "nin_shortcut.weight") in sd:
        x = conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def _ae_video_res_block(sd, p, x, T):
    """AE VideoResBlock.forward, autoencoding/temporal_ae.py:62-81: alpha * temporal + (1-alpha) * spatial."""
    x = _ae_resnet(sd, p, x)
    BT, C, H, W = x.shape
    x5 = x.view(BT // T, T, C, H, W).transpose(1, 2)
    xt = res_block(sd, p + "time_stack.", x5, None, dims=3)
    a = torch.sigmoid(sd[p + "mix_factor"])
    return (a * xt + (1.0 - a) * x5).transpose(1, 2).reshape(BT, C, H, W)


def _ae_attn(sd, p, x):
    """AttnBlock.forward, diffusionmodules/model.py:180-201 (one head of width C)."""
    B, C, H, W = x.shape
    h = _gn(sd, p + "norm", x, 1e-6)
    q, k, v = (conv2d(h, sd[p + n + ".weight"], sd[p + n + ".bias"]).flatten(2).transpose(1, 2) for n in ("q", "k", "v"))
    o = sdpa(q[:, None], k[:, None], v[:, None])[:, 0]
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return x + conv2d(o, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])


def vae_encoder(sd, cfg, x):
    """Encoder.forward, diffusionmodules/model.py:567-601 (Downsample :73-92: F.pad (0,1,0,1) + stride-2 conv, padding 0).
    x [n, 3, H, W] -> moments [n, 2 * z_channels, H/8, W/8]."""
    h = conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    for lvl in range(len(cfg.ch_mult)):
        for b in range(cfg.nrb):
            h = _ae_resnet(sd, f"down.{lvl}.block.{b}.", h)
        if lvl != len(cfg.ch_mult) - 1:
            h = conv2d(F.pad(h, (0, 1, 0, 1)), sd[f"down.{lvl}.downsample.conv.weight"], sd[f"down.{lvl}.downsample.conv.bias"], stride=2)
    h = _ae_resnet(sd, "mid.block_1.", h)
    h = _ae_attn(sd, "mid.attn_1.", h)
    h = _ae_resnet(sd, "mid.block_2.", h)
    h = F.silu(_gn(sd, "norm_out", h, 1e-6))
    return conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def vae_decoder_2d(sd, cfg, z):
    """Decoder.forward (2-D, no temporal layers), diffusionmodules/model.py:715-748; Upsample :52-70 (nearest 2x + conv)."""
    h = conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = _ae_resnet(sd, "mid.block_1.", h)
    h = _ae_attn(sd, "mid.attn_1.", h)
    h = _ae_resnet(sd, "mid.block_2.", h)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for b in range(cfg.nrb + 1):
            h = _ae_resnet(sd, f"up.{lvl}.block.{b}.", h)
        if lvl != 0:
            h = conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[f"up.{lvl}.upsample.conv.weight"],
                         sd[f"up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, "norm_out", h, 1e-6))
    return conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def cond_frame_encode(sd, cfg, x):
    """AutoencoderKLModeOnly.encode (sgm/models/autoencoder.py:468-490): encoder -> quant_conv -> mode (mean half of the moments)."""
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    m = conv2d(vae_encoder(enc, cfg, x), sd["quant_conv.weight"], sd["quant_conv.bias"])
    return m[:, : m.shape[1] // 2]


def video_decoder(sd, cfg, z, timesteps):
    """VideoDecoder / Decoder.forward, diffusionmodules/model.py:715-748 + temporal_ae.py:291-347, AE3DConv :99-105."""
    T = timesteps
    h = conv2d(z, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    h = _ae_video_res_block(sd, "mid.block_1.", h, T)
    h = _ae_attn(sd, "mid.attn_1.", h)
    h = _ae_video_res_block(sd, "mid.block_2.", h, T)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for b in range(cfg.nrb + 1):
            h = _ae_video_res_block(sd, f"up.{lvl}.block.{b}.", h, T)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv2d(h, sd[f"up.{lvl}.upsample.conv.weight"], sd[f"up.{lvl}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, "norm_out", h, 1e-6))
    h = conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    BT, C, H, W = h.shape
    h5 = h.view(BT // T, T, C, H, W).transpose(1, 2)
    h5 = conv3d(h5, sd["conv_out.time_mix_conv.weight"], sd["conv_out.time_mix_conv.bias"], padding=(1, 0, 0))
    return h5.transpose(1, 2).reshape(BT, C, H, W)


def decode_first_stage(sd, cfg, z, scale_factor=0.18215, max_chunk=8):
    """StreamingSVD.decode_first_stage, diffusion_trainer/streaming_svd.py:123-151 (groups of 8 frames, fp32)."""
    z = z / scale_factor
    outs = []
    for i in range(0, z.shape[0], max_chunk):
        zc = z[i:i + max_chunk]
        outs.append(video_decoder(sd, cfg, zc, zc.shape[0]))
    return torch.cat(outs, 0)


# ---- sampler ----------------------------------------------------------------------------------------------------
def ays_sigmas(n):
    """AlignYourSteps.get_sigmas + Discretization.__call__ (append zero), models/diffusion/discretizer.py:16-33,
    sgm discretizer.py:18-22.  float64."""
    sched = np.array([700.00, 54.5, 15.886, 7.977, 4.248, 1.789, 0.981, 0.403, 0.173, 0.034, 0.002])
    xs = np.linspace(0, 1, len(sched))
    ys = np.log(sched[::-1])
    new = np.exp(np.interp(np.linspace(0, 1, n), xs, ys))[::-1].copy()
    return torch.cat([torch.from_numpy(new), torch.zeros(1, dtype=torch.float64)])


def edm_sigmas(n, sigma_min=0.002, sigma_max=700.0, rho=7.0):
    """EDMDiscretization.get_sigmas + append zero, sgm discretizer.py:27-38 (fp32 ramp); the schedule of the first chunk."""
    ramp = torch.linspace(0, 1, n)
    mn, mx = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return torch.cat([(mx + ramp * (mn - mx)) ** rho, torch.zeros(1)])


def vscaling_edm(sigma):
    """VScalingWithEDMcNoise, denoiser_scaling.py:51-59 -> (c_skip, c_out, c_in, c_noise)."""
    return 1.0 / (sigma ** 2 + 1.0), -sigma / (sigma ** 2 + 1.0) ** 0.5, 1.0 / (sigma ** 2 + 1.0) ** 0.5, 0.25 * sigma.log()


def euler_edm_sample(network, x, cond, uc, num_steps, num_frames, min_scale=1.5, max_scale=3.0, sigmas=None):
    """EulerEDMSampler.__call__ (s_churn 0) + Denoiser.forward + LinearPredictionGuider,
    sampling.py:41-52,93-130 ; denoiser.py:23-39 ; guiders.py:60-99.
    network(x_in[2T..], c_noise[2T], cond_dict) -> eps-like output [2T..].
    sigmas: None = AlignYourSteps(num_steps) (the AR chunks); pass edm_sigmas(n) for the first chunk's Karras schedule."""
    sigmas = ays_sigmas(num_steps) if sigmas is None else sigmas.double()
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    x = x.float()
    s_in = x.new_ones([x.shape[0]])
    scale = torch.linspace(min_scale, max_scale, num_frames, device=x.device)
    for i in range(len(sigmas) - 1):
        sigma, nxt = s_in * sigmas[i], s_in * sigmas[i + 1]
        xin = torch.cat([x] * 2)
        s2 = torch.cat([sigma] * 2)
        c = {k: torch.cat((uc[k], cond[k]), 0) for k in ("vector", "crossattn", "concat")}
        sg = s2[:, None, None, None]
        c_skip, c_out, c_in, c_noise = vscaling_edm(sg)
        den = network(xin * c_in, c_noise.reshape(-1), c) * c_out + xin * c_skip
        xu, xc = den.chunk(2)
        den = xu + scale[:, None, None, None] * (xc - xu)
        d = (x - den) / sigma[:, None, None, None]
        x = x + d * (nxt - sigma)[:, None, None, None]
    return x
