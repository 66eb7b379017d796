"""Import bootstrap for the reference's vendored I2VGen-XL enhancer UNet (oracle tooling, NOT product code).

TEST INFRASTRUCTURE ONLY -- used by ``oracle/make_golden_i2v.py`` in the build container.  The enhancer's wiring lives in the
reference itself (code/i2v_enhance/unet_i2vgen_xl.py, unet_3d_blocks.py, transformer_2d.py, transformer_temporal.py,
attention.py) but its leaf layers are imported from **diffusers==0.30.2** (requirements.txt:6), which is absent from
/root/reference and from this image.  ``install()`` provides a fake ``diffusers`` package whose leaf classes restate the
published 0.30.2 behaviour (same constructor arguments, same parameter names, same arithmetic):

    Attention + AttnProcessor2_0, GEGLU, GELU, Timesteps / get_timestep_embedding, TimestepEmbedding, ResnetBlock2D,
    TemporalConvLayer, Downsample2D, Upsample2D, get_activation, ConfigMixin / register_to_config, ModelMixin.

With it the UNMODIFIED vendored ``I2VGenXLUNet`` imports and runs on CPU, which pins the *wiring* of oracle/i2vgen_oracle.py.
The leaf restatements themselves have no reference-side golden vectors: **parity unpinned** for them.
"""
import functools
import inspect
import math
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference/code"


# ------------------------------------------------------------------------------------------------ config / model mixins
class _Config(dict):
    __getattr__ = dict.__getitem__


class ConfigMixin:
    def register_to_config(self, **kw):
        if not hasattr(self, "_internal_dict"):
            object.__setattr__(self, "_internal_dict", _Config())
        self._internal_dict.update(kw)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    """diffusers.configuration_utils.register_to_config: record every init argument (with defaults) BEFORE running init."""

    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        params = [(n, p.default) for i, (n, p) in enumerate(inspect.signature(init).parameters.items()) if i > 0]
        cfg = {n: a for a, (n, _) in zip(args, params)}
        for n, d in params:
            if n not in cfg:
                cfg[n] = kwargs.get(n, d)
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return inner


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


# ------------------------------------------------------------------------------------------------ leaf layers (diffusers 0.30.2)
def get_activation(name):
    return {"silu": nn.SiLU, "swish": nn.SiLU, "gelu": nn.GELU, "relu": nn.ReLU, "mish": nn.Mish}[name.lower()]()


class GELU(nn.Module):
    """diffusers.models.activations.GELU: Linear then gelu (exact erf unless approximate='tanh')."""

    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(self.proj(x), approximate=self.approximate)


class GEGLU(nn.Module):
    """diffusers.models.activations.GEGLU: proj to 2*dim_out, value * gelu(gate)."""

    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, x, *a, **k):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention with AttnProcessor2_0 (the default on torch >= 2): no norm, no
    residual connection, rescale_output_factor 1, scale dim_head**-0.5."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, out_bias=True, **kw):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        x = hidden_states
        c = x if encoder_hidden_states is None else encoder_hidden_states
        B, N, _ = x.shape
        h = self.heads
        q, k, v = (p(t).view(B, t.shape[1], h, -1).transpose(1, 2)
                   for p, t in ((self.to_q, x), (self.to_k, c), (self.to_v, c)))
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(B, N, -1)
        return self.to_out[1](self.to_out[0](o))


def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1.0, scale=1.0,
                           max_period=10000):
    half = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift, scale=1):
        super().__init__()
        self.num_channels, self.flip, self.shift, self.scale = num_channels, flip_sin_to_cos, downscale_freq_shift, scale

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, self.flip, self.shift, self.scale)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, **kw):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = get_activation(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample, condition=None):
        return self.linear_2(self.act(self.linear_1(sample)))


class ResnetBlock2D(nn.Module):
    """diffusers.models.resnet.ResnetBlock2D, time_embedding_norm='default', no up/down."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512, groups=32,
                 groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", time_embedding_norm="default",
                 output_scale_factor=1.0, **kw):
        super().__init__()
        assert time_embedding_norm == "default"
        out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups_out or groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.output_scale_factor = output_scale_factor
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None, *a, **k):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class TemporalConvLayer(nn.Module):
    """diffusers.models.resnet.TemporalConvLayer: 4 x [GN, SiLU, (Dropout,) Conv3d (3,1,1)] + identity; conv4 zero-init."""

    def __init__(self, in_dim, out_dim=None, dropout=0.0, norm_num_groups=32):
        super().__init__()
        out_dim = out_dim or in_dim
        self.conv1 = nn.Sequential(nn.GroupNorm(norm_num_groups, in_dim), nn.SiLU(),
                                   nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        mk = lambda: nn.Sequential(nn.GroupNorm(norm_num_groups, out_dim), nn.SiLU(), nn.Dropout(dropout),
                                   nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        self.conv2, self.conv3, self.conv4 = mk(), mk(), mk()
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, hidden_states, num_frames=1):
        x = hidden_states[None, :].reshape((-1, num_frames) + hidden_states.shape[1:]).permute(0, 2, 1, 3, 4)
        identity = x
        x = self.conv4(self.conv3(self.conv2(self.conv1(x))))
        x = identity + x
        return x.permute(0, 2, 1, 3, 4).reshape((x.shape[0] * x.shape[2], -1) + x.shape[3:])


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", **kw):
        super().__init__()
        assert use_conv
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x, *a, **k):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv", **kw):
        super().__init__()
        assert use_conv and not use_conv_transpose
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None, *a, **k):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest") if output_size is None else \
            F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class _Unused:
    """Placeholder for names the vendored files import but the I2VGen-XL path never instantiates."""

    def __init__(self, *a, **k):
        raise NotImplementedError("not on the I2VGen-XL enhancer path")


def _stub(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ident = lambda f=None, *a, **k: f
    _stub("diffusers")
    _stub("diffusers.configuration_utils", ConfigMixin=ConfigMixin, LegacyConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _stub("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    class BaseOutput(dict):
        pass
    _stub("diffusers.utils", logging=_Logging, deprecate=lambda *a, **k: None, is_torch_version=lambda *a: True, BaseOutput=BaseOutput)
    _stub("diffusers.utils.torch_utils", apply_freeu=lambda *a, **k: (a[1], a[2]), maybe_allow_in_graph=ident)
    _stub("diffusers.models")
    _stub("diffusers.models.activations", get_activation=get_activation, GEGLU=GEGLU, GELU=GELU, ApproximateGELU=_Unused,
          FP32SiLU=_Unused, SwiGLU=_Unused)
    _stub("diffusers.models.attention_processor", Attention=Attention, JointAttnProcessor2_0=_Unused,
          ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=(), AttentionProcessor=object,
          AttnAddedKVProcessor=_Unused, AttnProcessor=_Unused, FusedAttnProcessor2_0=_Unused)
    _stub("diffusers.models.embeddings", TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps, ImagePositionalEmbeddings=_Unused,
          PatchEmbed=_Unused, PixArtAlphaTextProjection=_Unused, SinusoidalPositionalEmbedding=_Unused)
    _stub("diffusers.models.modeling_utils", ModelMixin=ModelMixin, LegacyModelMixin=ModelMixin)
    _stub("diffusers.models.modeling_outputs", Transformer2DModelOutput=BaseOutput)
    _stub("diffusers.models.normalization", AdaLayerNorm=_Unused, AdaLayerNormContinuous=_Unused, AdaLayerNormZero=_Unused,
          AdaLayerNormSingle=_Unused, RMSNorm=_Unused)
    _stub("diffusers.models.resnet", Downsample2D=Downsample2D, ResnetBlock2D=ResnetBlock2D, TemporalConvLayer=TemporalConvLayer,
          Upsample2D=Upsample2D, AlphaBlender=_Unused, SpatioTemporalResBlock=_Unused)
    _stub("diffusers.models.unets")
    _stub("diffusers.models.unets.unet_3d_condition", UNet3DConditionOutput=BaseOutput)
    _stub("diffusers.models.unets.unet_motion_model", **{n: type(n, (nn.Module,), {}) for n in
          ("DownBlockMotion", "CrossAttnDownBlockMotion", "UpBlockMotion", "CrossAttnUpBlockMotion", "UNetMidBlockCrossAttnMotion")})
