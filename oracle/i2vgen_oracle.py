"""fp32 CPU restatement of the reference's I2VGen-XL enhancer UNet and its denoising step.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product
path (streamingt2v_amd/).  Plain functional PyTorch on a state_dict that uses the reference's parameter names, so the real
``i2vgen-xl`` UNet weights would load unchanged.

What is pinned and what is not
  * WIRING (which layer feeds which, tensor layouts, reshapes, residuals, skip bookkeeping) follows the vendored reference
    code and is checked against it: oracle/make_golden_i2v.py runs the UNMODIFIED /root/reference/code/i2v_enhance modules
    (imported through oracle/i2v_bootstrap.py) on the same weights and inputs and requires agreement <= 2e-4.
      I2VGenXLUNet.forward                      code/i2v_enhance/unet_i2vgen_xl.py:573-814  (constructor :188-377)
      CrossAttnDownBlock3D / DownBlock3D        code/i2v_enhance/unet_3d_blocks.py:509-548, 616-636
      UNetMidBlock3DCrossAttn                   code/i2v_enhance/unet_3d_blocks.py:376-405
      CrossAttnUpBlock3D / UpBlock3D            code/i2v_enhance/unet_3d_blocks.py:734-795, 858-899
      Transformer2DModel (continuous, linear)   code/i2v_enhance/transformer_2d.py:479-492, 514-527
      TransformerTemporalModel                  code/i2v_enhance/transformer_temporal.py:121-200
      BasicTransformerBlock / FeedForward       code/i2v_enhance/attention.py:414-534, 1100-1158
      I2VGenXLTransformerTemporalEncoder        code/i2v_enhance/unet_i2vgen_xl.py:110-160
  * LEAF LAYERS come from diffusers==0.30.2 (requirements.txt:6), which is not vendored and not installed: Attention +
    AttnProcessor2_0, GEGLU, GELU, get_timestep_embedding / TimestepEmbedding, ResnetBlock2D, TemporalConvLayer,
    Downsample2D, Upsample2D, DDIMScheduler.step / add_noise.  They are restated here from the published algorithms.
    The reference holds no golden vectors for them  ==>  **parity unpinned** for the leaves.
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ leaves (diffusers 0.30.2)
def lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def conv2d(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def gn(sd, p, x, eps, groups=32):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def timestep_embedding(t, dim, flip_sin_to_cos=True, shift=0.0):
    """diffusers get_timestep_embedding as used by Timesteps(C, True, 0): [cos | sin], exponent / half."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - shift))
    arg = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], -1)
    return torch.cat([emb[:, half:], emb[:, :half]], -1) if flip_sin_to_cos else emb


def attention(sd, p, x, ctx, heads):
    """Attention + AttnProcessor2_0: softmax(q k^T / sqrt(d)) v, to_out[0]; q/k/v without bias in this model."""
    c = x if ctx is None else ctx
    B, N, _ = x.shape
    q, k, v = lin(sd, p + ".to_q", x), lin(sd, p + ".to_k", c), lin(sd, p + ".to_v", c)
    sp = lambda t: t.view(B, t.shape[1], heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
    return lin(sd, p + ".to_out.0", o.transpose(1, 2).reshape(B, N, -1))


def feed_forward_geglu(sd, p, x):
    h, gate = lin(sd, p + ".net.0.proj", x).chunk(2, -1)
    return lin(sd, p + ".net.2", h * F.gelu(gate))


def basic_block(sd, p, x, ctx, heads, double_self):
    """BasicTransformerBlock, norm_type 'layer_norm' (attention.py:414-534): LN-attn1-res, LN-attn2-res, LN-FF-res."""
    x = attention(sd, p + ".attn1", ln(sd, p + ".norm1", x), None, heads) + x
    x = attention(sd, p + ".attn2", ln(sd, p + ".norm2", x), None if double_self else ctx, heads) + x
    return feed_forward_geglu(sd, p + ".ff", ln(sd, p + ".norm3", x)) + x


def resnet(sd, p, x, temb, eps=1e-5):
    """ResnetBlock2D (default time embedding norm, output_scale_factor 1)."""
    h = conv2d(sd, p + ".conv1", F.silu(gn(sd, p + ".norm1", x, eps)))
    h = h + lin(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = conv2d(sd, p + ".conv2", F.silu(gn(sd, p + ".norm2", h, eps)))
    if p + ".conv_shortcut.weight" in sd:
        x = conv2d(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def temporal_conv_layer(sd, p, x, num_frames):
    """TemporalConvLayer: (b f) c h w -> b c f h w; 4 x [GN(eps 1e-5, stats over c/32*f*h*w), SiLU, Conv3d (3,1,1)]; + identity."""
    x = x[None, :].reshape((-1, num_frames) + x.shape[1:]).permute(0, 2, 1, 3, 4)
    identity = x
    for name, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        x = F.silu(gn(sd, f"{p}.{name}.0", x, 1e-5))
        x = F.conv3d(x, sd[f"{p}.{name}.{ci}.weight"], sd[f"{p}.{name}.{ci}.bias"], padding=(1, 0, 0))
    x = identity + x
    return x.permute(0, 2, 1, 3, 4).reshape((x.shape[0] * x.shape[2], -1) + x.shape[3:])


# ------------------------------------------------------------------------------------------------ vendored wiring
def transformer_2d(sd, p, x, ctx, head_dim=64):
    """Transformer2DModel, continuous input + linear projections (transformer_2d.py:479-492, 514-527).  x: (b f) c h w."""
    B, C, H, W = x.shape
    h = gn(sd, p + ".norm", x, 1e-6).permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = lin(sd, p + ".proj_in", h)
    h = basic_block(sd, p + ".transformer_blocks.0", h, ctx, h.shape[-1] // head_dim, double_self=False)
    h = lin(sd, p + ".proj_out", h).reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + x


def transformer_temporal(sd, p, x, num_frames, heads):
    """TransformerTemporalModel (transformer_temporal.py:160-195): GN over (c/32, f, h, w), tokens (b h w) f c, two self-attns."""
    BF, C, H, W = x.shape
    B = BF // num_frames
    h = x[None, :].reshape(B, num_frames, C, H, W).permute(0, 2, 1, 3, 4)
    h = gn(sd, p + ".norm", h, 1e-6)
    h = h.permute(0, 3, 4, 2, 1).reshape(B * H * W, num_frames, C)
    h = lin(sd, p + ".proj_in", h)
    h = basic_block(sd, p + ".transformer_blocks.0", h, None, heads, double_self=True)
    h = lin(sd, p + ".proj_out", h)
    h = h[None, None, :].reshape(B, H, W, num_frames, C).permute(0, 3, 4, 1, 2).reshape(BF, C, H, W)
    return h + x


def image_temporal_encoder(sd, p, x):
    """I2VGenXLTransformerTemporalEncoder (unet_i2vgen_xl.py:110-160): LN - attn(2 heads x 4) - res, GELU-FF - res (no norm)."""
    x = attention(sd, p + ".attn1", ln(sd, p + ".norm1", x), None, 2) + x
    return lin(sd, p + ".ff.net.2", F.gelu(lin(sd, p + ".ff.net.0.proj", x))) + x


def _count(sd, prefix):
    n = 0
    while f"{prefix}.{n}.norm1.weight" in sd or f"{prefix}.{n}.norm.weight" in sd or f"{prefix}.{n}.conv1.0.weight" in sd:
        n += 1
    return n


def constants(sd, image_latents, image_embeddings, text, fps, in_channels=4):
    """Everything in I2VGenXLUNet.forward that does not depend on the sample or the timestep (unet_i2vgen_xl.py:655-712):
    fps embedding, the 77 + 64 + 4 context tokens, and the processed image latents that are concatenated to the sample."""
    B, C, Fr, H, W = image_latents.shape
    cdim = text.shape[-1]
    model_ch = sd["conv_in.weight"].shape[0]
    fps_emb = lin(sd, "fps_embedding.2", F.silu(lin(sd, "fps_embedding.0", timestep_embedding(fps, model_ch))))
    first = image_latents[:, :, :1].permute(0, 2, 1, 3, 4).reshape(B, C, H, W)
    c = F.silu(conv2d(sd, "image_latents_context_embedding.0", first))
    c = F.adaptive_avg_pool2d(c, (32, 32))
    c = F.silu(conv2d(sd, "image_latents_context_embedding.3", c, stride=2))
    c = conv2d(sd, "image_latents_context_embedding.5", c, stride=2)
    c = c.permute(0, 2, 3, 1).reshape(B, -1, c.shape[1])
    img = lin(sd, "context_embedding.2", F.silu(lin(sd, "context_embedding.0", image_embeddings))).view(-1, in_channels, cdim)
    context = torch.cat([text, c, img], 1)                                      # [B, 77 + 64 + 4, cdim]
    il = image_latents.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, H, W)
    il = F.silu(conv2d(sd, "image_latents_proj_in.0", il))
    il = F.silu(conv2d(sd, "image_latents_proj_in.2", il))
    il = conv2d(sd, "image_latents_proj_in.4", il)
    il = il[None, :].reshape(B, Fr, C, H, W).permute(0, 3, 4, 1, 2).reshape(B * H * W, Fr, C)
    il = image_temporal_encoder(sd, "image_latents_temporal_encoder", il)
    il = il.reshape(B, H, W, Fr, C).permute(0, 4, 3, 1, 2)                      # [B, 4, F, H, W]
    return fps_emb, context, il


def unet(sd, sample, t, fps, image_latents, image_embeddings, text, head_dim=64):
    """I2VGenXLUNet.forward (unet_i2vgen_xl.py:573-814).  sample [B,4,F,h,w]; t scalar tensor; fps [B]; image_latents
    [B,4,F,h,w]; image_embeddings [B,cdim]; text [B,77,cdim]  ->  [B,4,F,h,w]."""
    B, C, Fr, H, W = sample.shape
    model_ch = sd["conv_in.weight"].shape[0]
    fps_emb, context, il = constants(sd, image_latents, image_embeddings, text, fps, C)
    tt = t.reshape(-1).expand(B)
    t_emb = lin(sd, "time_embedding.linear_2", F.silu(lin(sd, "time_embedding.linear_1", timestep_embedding(tt, model_ch))))
    emb = (t_emb + fps_emb).repeat_interleave(Fr, 0)
    ctx = context.repeat_interleave(Fr, 0)

    x = torch.cat([sample, il], 1).permute(0, 2, 1, 3, 4).reshape(B * Fr, 2 * C, H, W)
    x = conv2d(sd, "conv_in", x)
    x = transformer_temporal(sd, "transformer_in", x, Fr, heads=8)

    n_down = 0
    while f"down_blocks.{n_down}.resnets.0.norm1.weight" in sd:
        n_down += 1
    n_upsamplers = sum(1 for i in range(n_down) if f"up_blocks.{i}.upsamplers.0.conv.weight" in sd)
    forward_size = any(s % (2 ** n_upsamplers) != 0 for s in (H, W))

    skips = [x]
    for i in range(n_down):
        p = f"down_blocks.{i}"
        has_attn = f"{p}.attentions.0.norm.weight" in sd
        for j in range(_count(sd, p + ".resnets")):
            x = resnet(sd, f"{p}.resnets.{j}", x, emb)
            x = temporal_conv_layer(sd, f"{p}.temp_convs.{j}", x, Fr)
            if has_attn:
                x = transformer_2d(sd, f"{p}.attentions.{j}", x, ctx, head_dim)
                x = transformer_temporal(sd, f"{p}.temp_attentions.{j}", x, Fr, x.shape[1] // head_dim)
            skips.append(x)
        if f"{p}.downsamplers.0.conv.weight" in sd:
            x = conv2d(sd, f"{p}.downsamplers.0.conv", x, stride=2)
            skips.append(x)

    p = "mid_block"
    x = resnet(sd, p + ".resnets.0", x, emb)
    x = temporal_conv_layer(sd, p + ".temp_convs.0", x, Fr)
    x = transformer_2d(sd, p + ".attentions.0", x, ctx, head_dim)
    x = transformer_temporal(sd, p + ".temp_attentions.0", x, Fr, x.shape[1] // head_dim)
    x = resnet(sd, p + ".resnets.1", x, emb)
    x = temporal_conv_layer(sd, p + ".temp_convs.1", x, Fr)

    for i in range(n_down):
        p = f"up_blocks.{i}"
        has_attn = f"{p}.attentions.0.norm.weight" in sd
        n = _count(sd, p + ".resnets")
        res, skips = skips[-n:], skips[:-n]
        for j in range(n):
            x = torch.cat([x, res[-1 - j]], 1)
            x = resnet(sd, f"{p}.resnets.{j}", x, emb)
            x = temporal_conv_layer(sd, f"{p}.temp_convs.{j}", x, Fr)
            if has_attn:
                x = transformer_2d(sd, f"{p}.attentions.{j}", x, ctx, head_dim)
                x = transformer_temporal(sd, f"{p}.temp_attentions.{j}", x, Fr, x.shape[1] // head_dim)
        if f"{p}.upsamplers.0.conv.weight" in sd:
            size = skips[-1].shape[2:] if forward_size else None
            x = F.interpolate(x, scale_factor=2.0, mode="nearest") if size is None else F.interpolate(x, size=size, mode="nearest")
            x = conv2d(sd, f"{p}.upsamplers.0.conv", x)

    x = conv2d(sd, "conv_out", F.silu(gn(sd, "conv_norm_out", x, 1e-5)))
    return x[None, :].reshape((-1, Fr) + x.shape[1:]).permute(0, 2, 1, 3, 4)


# ------------------------------------------------------------------------------------------------ DDIM (diffusers 0.30.2)
class DDIM:
    """DDIMScheduler as configured by ali-vilab/i2vgen-xl (scheduler_config.json: scaled_linear betas 0.00085..0.012, 1000
    train steps, clip_sample False, set_alpha_to_one False -> final_alpha_cumprod = alphas_cumprod[0], steps_offset 1,
    timestep_spacing 'leading', prediction_type 'v_prediction', rescale_betas_zero_snr True) -- restated; parity unpinned.
    Only what pipeline_i2vgen_xl.py:812-816, 541-551, 605-613, 884-885 uses: set_timesteps, add_noise, step (eta 0)."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, prediction_type="v_prediction",
                 rescale_betas_zero_snr=True, set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        if rescale_betas_zero_snr:
            ac = torch.cumprod(1.0 - betas, 0).sqrt()
            a0, aT = ac[0].clone(), ac[-1].clone()
            ac = (ac - aT) * a0 / (a0 - aT)
            ab = ac ** 2
            alphas = torch.cat([ab[0:1], ab[1:] / ab[:-1]])
            betas = 1 - alphas
        self.alphas_cumprod = torch.cumprod(1.0 - betas, 0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train, self.offset, self.pred = num_train, steps_offset, prediction_type

    def set_timesteps(self, n):
        self.n = n
        ratio = self.num_train // n
        self.timesteps = (torch.arange(0, n) * ratio).round().flip(0).long() + self.offset

    def add_noise(self, x0, noise, t):
        a = self.alphas_cumprod[t]
        return a.sqrt() * x0 + (1 - a).sqrt() * noise

    def step(self, model_out, t, sample):
        prev_t = t - self.num_train // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        if self.pred == "v_prediction":
            x0 = a_t.sqrt() * sample - b_t.sqrt() * model_out
            eps = a_t.sqrt() * model_out + b_t.sqrt() * sample
        else:
            x0 = (sample - b_t.sqrt() * model_out) / a_t.sqrt()
            eps = model_out
        return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps


# ------------------------------------------------------------------------------------------------ the pipeline call
def center_crop_wide(img, resolution):
    """pipeline_i2vgen_xl.py:965-1000 (_center_crop_wide) for one PIL image."""
    import PIL.Image
    scale = min(img.size[0] / resolution[0], img.size[1] / resolution[1])
    img = img.resize((round(img.width // scale), round(img.height // scale)), resample=PIL.Image.BOX)
    x1, y1 = (img.width - resolution[0]) // 2, (img.height - resolution[1]) // 2
    return img.crop((x1, y1, x1 + resolution[0], y1 + resolution[1]))


def enhance_call(sd, images, frames, prompt_embeds, negative_prompt_embeds, vae, image_encoder, generator, py_random, *, height, width, chunk_size,
                 overlap_size, num_inference_steps, strength, guidance_scale, target_fps=38, clip_mean=(0.48145466, 0.4578275, 0.40821073),
                 clip_std=(0.26862954, 0.26130258, 0.27577711), crop_size=224, trace=None):
    """I2VGenXLPipeline.__call__ (pipeline_i2vgen_xl.py:730-925) with output_type='latent', in the order the reference consumes its random
    streams: per key image [CLIP embedding, then after all embeddings: VAE posterior sample of the wide crop], the video's posterior sample,
    the SDEdit noise; one `random.randint` per blending window after the first, per DDIM step.
    vae: .encode(x).latent_dist.sample(generator), .config.scaling_factor;  image_encoder(pixels).image_embeds.
    trace (optional dict) receives the intermediates (conds per window, initial latents, timesteps)."""
    import numpy as np
    import PIL.Image
    to_pt = lambda pil: torch.from_numpy(np.stack([np.array(p).astype(np.float32) / 255.0 for p in pil], 0).transpose(0, 3, 1, 2))
    text = torch.cat([negative_prompt_embeds, prompt_embeds])                                     # :764-765  [uncond | cond]
    mean, std = torch.tensor(clip_mean).view(1, 3, 1, 1), torch.tensor(clip_std).view(1, 3, 1, 1)
    embs = []
    for img in images:                                                                             # :771-780, _encode_image :349-383
        sq = center_crop_wide(img, (width, width)).resize((crop_size, crop_size), PIL.Image.BILINEAR)
        e = image_encoder((to_pt([sq]) - mean) / std).image_embeds.unsqueeze(1)
        embs.append(torch.cat([torch.zeros_like(e), e]))
    n_win = len(images)
    lats = []
    for img in images:                                                                             # :784-795, prepare_image_latents :479-511
        wide = center_crop_wide(img, (width, height))
        il = vae.encode(2.0 * to_pt([wide]) - 1.0).latent_dist.sample() * vae.config.scaling_factor   # NB: no generator (:486)
        il = il.unsqueeze(2)
        planes = [torch.ones_like(il[:, :, :1]) * ((i + 1) / (chunk_size - 1)) for i in range(chunk_size - 1)]
        il = torch.cat([il] + planes, 2) if planes else il
        lats.append(torch.cat([il] * 2))
    fps = torch.tensor([target_fps, target_fps])
    video = 2.0 * to_pt([PIL.Image.fromarray(np.asarray(f)) for f in frames]) - 1.0               # [F, 3, H, W]
    sched = DDIM()
    sched.set_timesteps(num_inference_steps)
    t_start = max(num_inference_steps - min(int(num_inference_steps * strength), num_inference_steps), 0)   # get_timesteps :541-551
    ts = sched.timesteps.tolist()[t_start:]
    F_ = video.shape[0]
    if F_ > 16:                                                                                    # prepare_video_latents :585-597
        init = torch.cat([vae.encode(ch).latent_dist.sample(generator) for ch in torch.chunk(video, F_ // 16, 0)], 0)
    else:
        init = vae.encode(video).latent_dist.sample(generator)
    init = vae.config.scaling_factor * init
    noise = torch.randn(init.shape, generator=generator)
    lat = sched.add_noise(init, noise, ts[0])
    lat = lat[None].permute(0, 2, 1, 3, 4)                                                         # [1, 4, F, h, w]
    if trace is not None:
        b5 = lambda x: x[None].permute(0, 2, 1, 3, 4).clone()
        trace.update(image_embeddings=embs, image_latents=lats, fps=fps, text=text, init_latents=lat.clone(), timesteps=ts, clean=b5(init), noise=b5(noise))
    for t in ts:                                                                                   # :841-905
        out = torch.empty_like(lat)
        start = 0
        for idx in range(n_win):
            w = lat[:, :, start:start + chunk_size]
            pred = unet(sd, torch.cat([w] * 2), torch.tensor(t), fps, lats[idx], embs[idx].squeeze(1), text)
            pu, pc = pred.chunk(2)
            pred = pu + guidance_scale * (pc - pu)
            B, C, Fr, h, w_ = w.shape
            fr = lambda x: x.permute(0, 2, 1, 3, 4).reshape(B * Fr, C, h, w_)
            new = sched.step(fr(pred), t, fr(w))[None].reshape(B, Fr, C, h, w_).permute(0, 2, 1, 3, 4)
            off = 0 if start == 0 or overlap_size == 0 else py_random.randint(0, overlap_size - 1)
            out[:, :, start + off:start + chunk_size] = new[:, :, off:]
            start += chunk_size - overlap_size
        lat = out
        if start + overlap_size > lat.shape[2]:
            raise NotImplementedError("video does not divide into chunks")
    return lat
