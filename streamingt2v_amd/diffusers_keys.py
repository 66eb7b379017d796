"""Key maps between the diffusers layout of stock SVD-XT and the sgm layout the kernels' modules are specified in.

Why: the reference generates CHUNK 0 with diffusers' ``StableVideoDiffusionPipeline.from_pretrained("stabilityai/stable-video-diffusion-
img2vid-xt", torch_dtype=float16, variant="fp16")`` (code/config.yaml:280-299, code/diffusion_trainer/streaming_svd.py:388-390): the first 25
frames come from the STOCK weights, stored under diffusers' names (``unet/`` = UNetSpatioTemporalConditionModel, ``vae/`` = AutoencoderKL-
TemporalDecoder, ``image_encoder/`` = CLIPVisionModelWithProjection).  UNetSpatioTemporalConditionModel is the re-keyed sgm VideoUNet (the
network of code/models/diffusion/video_model.py:88 without ControlNet / CAM) and AutoencoderKLTemporalDecoder the re-keyed sgm
AutoencodingEngine with VideoDecoder -- same tensors, same shapes, other names -- so ``pipeline.load_stock_svd_xt`` loads that folder into
VideoUNet(controlnet_mode=False) / VideoDecoder / CondFrameEncoder through the maps below.

diffusers is neither vendored by the reference nor installed here (requirements.txt:6 pins diffusers==0.30.2): the names are restated from
the published module definitions (models/unets/unet_spatio_temporal_condition.py, unet_3d_blocks.py, transformers/transformer_temporal.py,
autoencoders/autoencoder_kl_temporal_decoder.py) = the inverse of diffusers' scripts/convert_svd_to_diffusers.py.  **Parity unpinned** against
the real files; what IS checked (tests/test_host_logic.py): every tensor of our specs gets exactly one diffusers name and back (bijection,
strict both ways), shapes carry over, and the totals are the published ones (UNet 1 524 623 082 parameters, VAE 97 742 847).
"""

_RES = {"in_layers.0.": "norm1.", "in_layers.2.": "conv1.", "emb_layers.1.": "time_emb_proj.", "out_layers.0.": "norm2.", "out_layers.3.": "conv2.",
        "skip_connection.": "conv_shortcut."}


def _res(rest):
    """VideoResBlock (openaimodel.ResBlock + time_stack + AlphaBlender) -> SpatioTemporalResBlock."""
    if rest.startswith("time_mixer."):
        return rest
    sub = "spatial_res_block."
    if rest.startswith("time_stack."):
        sub, rest = "temporal_res_block.", rest[len("time_stack."):]
    for a, b in _RES.items():
        if rest.startswith(a):
            return sub + b + rest[len(a):]
    raise KeyError(rest)


def _attn(rest):
    """SpatialVideoTransformer -> TransformerSpatioTemporalModel."""
    if rest.startswith("time_stack.0."):
        return "temporal_transformer_blocks.0." + rest[len("time_stack.0."):]
    if rest.startswith("time_pos_embed."):
        i, tail = rest[len("time_pos_embed."):].split(".", 1)
        return f"time_pos_embed.linear_{ {'0': 1, '2': 2}[i] }.{tail}"
    return rest          # norm, proj_in, proj_out, transformer_blocks.0.*, time_mixer.mix_factor


def sgm_unet_key_to_diffusers(name, num_res_blocks=2):
    """One VideoUNet parameter name (sgm) -> its UNetSpatioTemporalConditionModel name."""
    per = num_res_blocks + 1
    head, _, rest = name.partition(".")
    if head == "time_embed":
        i, tail = rest.split(".", 1)
        return f"time_embedding.linear_{ {'0': 1, '2': 2}[i] }.{tail}"
    if head == "label_emb":
        z, i, tail = rest.split(".", 2)
        assert z == "0"
        return f"add_embedding.linear_{ {'0': 1, '2': 2}[i] }.{tail}"
    if head == "out":
        i, tail = rest.split(".", 1)
        return {"0": "conv_norm_out.", "2": "conv_out."}[i] + tail
    if head == "input_blocks":
        i, m, tail = rest.split(".", 2)
        i = int(i)
        if i == 0:
            return "conv_in." + tail
        b, j = (i - 1) // per, (i - 1) % per
        if j == num_res_blocks:                                   # Downsample: input_blocks.{i}.0.op
            assert m == "0" and tail.startswith("op.")
            return f"down_blocks.{b}.downsamplers.0.conv.{tail[3:]}"
        return f"down_blocks.{b}.resnets.{j}.{_res(tail)}" if m == "0" else f"down_blocks.{b}.attentions.{j}.{_attn(tail)}"
    if head == "middle_block":
        m, tail = rest.split(".", 1)
        if m == "1":
            return "mid_block.attentions.0." + _attn(tail)
        return f"mid_block.resnets.{ {'0': 0, '2': 1}[m] }." + _res(tail)
    if head == "output_blocks":
        i, m, tail = rest.split(".", 2)
        b, j = int(i) // per, int(i) % per
        if m == "0":
            return f"up_blocks.{b}.resnets.{j}.{_res(tail)}"
        if tail.startswith("conv."):                               # Upsample sits at .1 (level without attention) or .2
            return f"up_blocks.{b}.upsamplers.0.{tail}"
        return f"up_blocks.{b}.attentions.{j}.{_attn(tail)}"
    raise KeyError(name)


def svd_unet_diffusers_to_sgm(sd, spec, num_res_blocks=2, prefix=""):
    """diffusers UNetSpatioTemporalConditionModel state_dict -> {sgm name: tensor} for every entry of `spec` (VideoUNet(controlnet_mode=False)
    .spec()).  Strict: a missing tensor, a shape mismatch or an unconsumed diffusers key raises."""
    out, used = {}, set()
    for name, shape in spec:
        k = prefix + sgm_unet_key_to_diffusers(name, num_res_blocks)
        if k not in sd:
            raise KeyError(f"stock SVD UNet: no tensor {k!r} (for {name})")
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"stock SVD UNet: {k} has shape {tuple(sd[k].shape)}, {name} expects {tuple(shape)}")
        out[name] = sd[k]
        used.add(k)
    extra = [k for k in sd if k.startswith(prefix) and k not in used]
    if extra:
        raise KeyError(f"stock SVD UNet: {len(extra)} tensors without a counterpart, e.g. {extra[:4]}")
    return out


_VRES = {"norm1.": "spatial_res_block.norm1.", "conv1.": "spatial_res_block.conv1.", "norm2.": "spatial_res_block.norm2.", "conv2.": "spatial_res_block.conv2.",
         "nin_shortcut.": "spatial_res_block.conv_shortcut.", "time_stack.in_layers.0.": "temporal_res_block.norm1.", "time_stack.in_layers.2.": "temporal_res_block.conv1.",
         "time_stack.out_layers.0.": "temporal_res_block.norm2.", "time_stack.out_layers.3.": "temporal_res_block.conv2.", "mix_factor": "time_mixer.mix_factor"}
_VATT = {"norm.": "group_norm.", "q.": "to_q.", "k.": "to_k.", "v.": "to_v.", "proj_out.": "to_out.0."}


def _sub(rest, table):
    for a, b in table.items():
        if rest.startswith(a):
            return b + rest[len(a):]
    raise KeyError(rest)


def sgm_temporal_decoder_key_to_diffusers(name, n_levels=4):
    """One VideoDecoder parameter name (sgm, relative to ``first_stage_model.decoder.``) -> its name inside AutoencoderKLTemporalDecoder.
    Attention q / k / v / proj_out are 1x1 convolutions in sgm and Linear layers in diffusers (weights reshaped by the caller)."""
    if name.startswith("conv_out.time_mix_conv."):
        return "decoder.time_conv_out." + name[len("conv_out.time_mix_conv."):]
    if name.startswith(("conv_in.", "conv_out.")):
        return "decoder." + name
    if name.startswith("norm_out."):
        return "decoder.conv_norm_out." + name[len("norm_out."):]
    if name.startswith("mid.attn_1."):
        return "decoder.mid_block.attentions.0." + _sub(name[len("mid.attn_1."):], _VATT)
    if name.startswith("mid.block_"):
        i, tail = name[len("mid.block_"):].split(".", 1)
        return f"decoder.mid_block.resnets.{int(i) - 1}." + _sub(tail, _VRES)
    if name.startswith("up."):
        _, lvl, kind, tail = name.split(".", 3)
        b = n_levels - 1 - int(lvl)
        if kind == "upsample":
            return f"decoder.up_blocks.{b}.upsamplers.0.{tail}"
        j, tail = tail.split(".", 1)
        return f"decoder.up_blocks.{b}.resnets.{j}." + _sub(tail, _VRES)
    raise KeyError(name)


def svd_vae_diffusers_to_sgm(sd, decoder_spec, encoder_spec, n_levels=4):
    """diffusers AutoencoderKLTemporalDecoder state_dict -> ({VideoDecoder name: tensor}, {CondFrameEncoder name: tensor}).  The encoder half
    is diffusers' ordinary Encoder (temporal_ae.diffusers_vae_to_sgm_keys already maps it); strict like svd_unet_diffusers_to_sgm."""
    from .temporal_ae import diffusers_vae_to_sgm_keys
    dec, used = {}, set()
    for name, shape in decoder_spec:
        k = sgm_temporal_decoder_key_to_diffusers(name, n_levels)
        if k not in sd:
            raise KeyError(f"stock SVD VAE: no tensor {k!r} (for decoder.{name})")
        v = sd[k]
        if v.dim() == 2 and len(shape) == 4:
            v = v[:, :, None, None]
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"stock SVD VAE: {k} has shape {tuple(sd[k].shape)}, decoder.{name} expects {tuple(shape)}")
        dec[name] = v
        used.add(k)
    enc_sd = diffusers_vae_to_sgm_keys({k: v for k, v in sd.items() if not k.startswith("decoder.")}, n_levels)
    enc = {}
    for name, shape in encoder_spec:
        if name not in enc_sd:
            raise KeyError(f"stock SVD VAE: no tensor for {name}")
        if tuple(enc_sd[name].shape) != tuple(shape):
            raise ValueError(f"stock SVD VAE: {name} has shape {tuple(enc_sd[name].shape)}, expected {tuple(shape)}")
        enc[name] = enc_sd[name]
    extra = [k for k in sd if k.startswith("decoder.") and k not in used] + [k for k in enc_sd if k not in enc]
    if extra:
        raise KeyError(f"stock SVD VAE: {len(extra)} tensors without a counterpart, e.g. {extra[:4]}")
    return dec, enc
