"""OpenCLIP ViT-H/14 image tower on the HIP kernels: the `crossattn` conditioning of StreamingSVD is the CLIP
image embedding of the conditioning frame (FrozenOpenCLIPImageEmbedder.forward -> self.model.visual(img),
code/models/svd/sgm/modules/encoders/modules.py:574-732; config.yaml:160-175).

open_clip==2.24.0 is neither vendored nor installed: the tower is the published VisionTransformer of that package
(conv1 patch embedding without bias, class token, learned positions, ln_pre, 32 x ResidualAttentionBlock[ln_1, nn.MultiheadAttention,
ln_2, c_fc - GELU - c_proj], class-token pooling, ln_post, projection) -- **parity unpinned**.  state_dict keys are open_clip's
(``visual.*``; in the reference checkpoint under ``conditioner.embedders.0.open_clip.model.``).  Input: images already resized to
224 x 224 and normalised with the CLIP mean/std (the reference does that with kornia's antialiased bicubic resize, modules.py:624-636,
which is not reproduced here).

Head dim is 80 (1280 / 16), not the 64 of the flash-attention kernel, and the sequence is 257 tokens for ONE image per chunk: the
attention runs as per-head GEMMs (scores fp32 -> row softmax -> PV) with each head zero-padded to 96 channels and the sequence to 260
tokens (masked as keys through the score GEMM's bias vector).  ~1600 small launches once per 25-frame chunk.
"""
import torch

from . import ops
from .params import Spec, check_state_dict
from .video_model import _dev_bf16, _dev_f32


class ClipVisionConfig:
    def __init__(self, width=1280, layers=32, heads=16, patch_size=14, image_size=224, embed_dim=1024, mlp_ratio=4):
        self.width, self.layers, self.heads, self.patch, self.image, self.embed_dim = width, layers, heads, patch_size, image_size, embed_dim
        self.mlp = int(width * mlp_ratio)
        self.grid = image_size // patch_size
        self.hd = width // heads
        self.hd_pad = (self.hd + 31) // 32 * 32
        assert width % 8 == 0 and self.mlp % 8 == 0


class OpenCLIPVisionTower:
    def __init__(self, cfg=None, prefix="visual."):
        self.cfg, self.p = cfg or ClipVisionConfig(), prefix

    def spec(self):
        c, p, s = self.cfg, self.p, Spec()
        s.add(p + "class_embedding", c.width); s.add(p + "positional_embedding", c.grid * c.grid + 1, c.width)
        s.add(p + "proj", c.width, c.embed_dim); s.add(p + "conv1.weight", c.width, 3, c.patch, c.patch)
        for n in ("ln_pre", "ln_post"):
            s.add(p + n + ".weight", c.width); s.add(p + n + ".bias", c.width)
        for i in range(c.layers):
            b = f"{p}transformer.resblocks.{i}."
            for n in ("ln_1", "ln_2"):
                s.add(b + n + ".weight", c.width); s.add(b + n + ".bias", c.width)
            s.add(b + "attn.in_proj_weight", 3 * c.width, c.width); s.add(b + "attn.in_proj_bias", 3 * c.width)
            s.add(b + "attn.out_proj.weight", c.width, c.width); s.add(b + "attn.out_proj.bias", c.width)
            s.add(b + "mlp.c_fc.weight", c.mlp, c.width); s.add(b + "mlp.c_fc.bias", c.mlp)
            s.add(b + "mlp.c_proj.weight", c.width, c.mlp); s.add(b + "mlp.c_proj.bias", c.width)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        sd = {k: v for k, v in sd.items() if k.startswith(self.p)}
        check_state_dict(self.spec(), sd)
        c, p, dev = self.cfg, self.p, device
        g = lambda k: sd[p + k].detach().float()
        kp = c.patch * c.patch * 3
        self.kpad = (kp + 31) // 32 * 32
        w = torch.zeros(c.width, self.kpad)
        w[:, :kp] = g("conv1.weight").reshape(c.width, kp)                       # (c, ky, kx) order = the unfold order below
        self.w_patch = _dev_bf16(w, dev)
        self.cls, self.pos = g("class_embedding").to(dev), g("positional_embedding").to(dev)
        self.ln_pre, self.ln_post = (_dev_f32(g("ln_pre.weight"), dev), _dev_f32(g("ln_pre.bias"), dev)), \
                                    (_dev_f32(g("ln_post.weight"), dev), _dev_f32(g("ln_post.bias"), dev))
        self.w_proj = _dev_bf16(g("proj").t().contiguous(), dev)                 # x @ proj  ==  linear with weight proj^T
        H, hd, hp = c.heads, c.hd, c.hd_pad

        def pad_heads_rows(wm, bm):          # [H*hd, K] -> [H*hp, K]: every head followed by hp - hd zero rows
            wo = torch.zeros(H, hp, wm.shape[1]); wo[:, :hd] = wm.view(H, hd, -1)
            bo = torch.zeros(H, hp); bo[:, :hd] = bm.view(H, hd)
            return wo.view(H * hp, -1), bo.view(H * hp)

        self.blocks = []
        for i in range(c.layers):
            b = f"transformer.resblocks.{i}."
            wi, bi = g(b + "attn.in_proj_weight"), g(b + "attn.in_proj_bias")
            (wq, bq), (wk, bk), (wv, bv) = (pad_heads_rows(wi[j * c.width:(j + 1) * c.width], bi[j * c.width:(j + 1) * c.width]) for j in range(3))
            wo = torch.zeros(c.width, H, hp); wo[:, :, :hd] = g(b + "attn.out_proj.weight").view(c.width, H, hd)
            self.blocks.append(dict(
                ln1=(_dev_f32(g(b + "ln_1.weight"), dev), _dev_f32(g(b + "ln_1.bias"), dev)),
                ln2=(_dev_f32(g(b + "ln_2.weight"), dev), _dev_f32(g(b + "ln_2.bias"), dev)),
                wqk=_dev_bf16(torch.cat([wq, wk], 0), dev), bqk=_dev_f32(torch.cat([bq, bk], 0), dev),
                wv=_dev_bf16(wv, dev), bv=_dev_f32(bv, dev),
                wo=_dev_bf16(wo.view(c.width, H * hp), dev), bo=_dev_f32(g(b + "attn.out_proj.bias"), dev),
                w1=_dev_bf16(g(b + "mlp.c_fc.weight"), dev), b1=_dev_f32(g(b + "mlp.c_fc.bias"), dev),
                w2=_dev_bf16(g(b + "mlp.c_proj.weight"), dev), b2=_dev_f32(g(b + "mlp.c_proj.bias"), dev)))
        self.device = dev
        return self

    def forward(self, images):
        """images fp32 [B, 3, 224, 224], CLIP-normalised -> image embeddings fp32 [B, embed_dim]."""
        c, dev = self.cfg, self.device
        B, G, P = images.shape[0], c.grid, c.patch
        n_tok = G * G + 1
        T = (n_tok + 3) // 4 * 4                                  # sequence padded to a multiple of 4 rows (masked as keys)
        tld = (T + 31) // 32 * 32                                 # K of the PV GEMM
        H, hp, W = c.heads, c.hd_pad, c.width
        # patch embedding: unfold (plumbing) -> GEMM with the positional embedding as residual
        x = images.to(dev, torch.float32).view(B, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, 3 * P * P)
        xp = torch.zeros((B * G * G, self.kpad), dtype=torch.float32, device=dev)
        xp[:, : 3 * P * P] = x
        pos = ops.to_elem(self.pos[1:].repeat(B, 1).contiguous())
        patches = ops.gemm(ops.to_elem(xp), self.w_patch, residual=pos)
        tok = torch.zeros((B, T, W), dtype=patches.dtype, device=dev)
        tok[:, 1:n_tok] = patches.view(B, G * G, W)
        tok[:, 0] = ops.to_elem((self.cls + self.pos[0]).contiguous())
        x = ops.layernorm(tok.view(B * T, W), *self.ln_pre)
        mask = torch.zeros(T, dtype=torch.float32, device=dev)
        mask[n_tok:] = -1e30
        vt = torch.zeros((B, H * hp, tld), dtype=x.dtype, device=dev)
        s = torch.empty((T, T), dtype=torch.float32, device=dev)
        pm = torch.zeros((T, tld), dtype=x.dtype, device=dev)
        o = torch.empty((B * T, H * hp), dtype=x.dtype, device=dev)
        scale = c.hd ** -0.5
        for blk in self.blocks:
            n1 = ops.layernorm(x, *blk["ln1"])
            qk = ops.gemm(n1, blk["wqk"], bias=blk["bqk"])                                            # [B*T, 2*H*hp]
            ops.gemm(n1, blk["wv"], bias=blk["bv"], trans_out=dict(tok_per_frame=T, tokens_ld=tld, out=vt))
            for b in range(B):
                rows = slice(b * T, (b + 1) * T)
                for h in range(H):
                    ops.gemm(qk[rows, h * hp:(h + 1) * hp], qk[rows, (H + h) * hp:(H + h + 1) * hp], bias=mask, out=s)
                    ops.softmax_rows(s, pm[:, :T], scale)
                    ops.gemm(pm, vt[b, h * hp:(h + 1) * hp], out=o[rows, h * hp:(h + 1) * hp])
            x = ops.gemm(o, blk["wo"], bias=blk["bo"], residual=x)
            m = ops.gelu_(ops.gemm(ops.layernorm(x, *blk["ln2"]), blk["w1"], bias=blk["b1"]))
            x = ops.gemm(m, blk["w2"], bias=blk["b2"], residual=x)
        pooled = x.view(B, T, W)[:, 0].contiguous()                                                  # class token
        return ops.gemm(ops.layernorm(pooled, *self.ln_post), self.w_proj, out_f32=True)

    __call__ = forward


def hf_clip_vision_to_openclip_keys(hf_sd, layers, prefix=""):
    """transformers CLIPVisionModelWithProjection state_dict (``vision_model.*`` + ``visual_projection.weight``: the enhancer's
    image_encoder, pipeline_i2vgen_xl.py:349-383) -> the open_clip key names this tower is specified in.  Same architecture; the
    mapping is verified numerically in oracle/check_clip_vs_hf.py."""
    g = lambda k: hf_sd[prefix + "vision_model." + k]
    sd = {"visual.class_embedding": g("embeddings.class_embedding"), "visual.positional_embedding": g("embeddings.position_embedding.weight"),
          "visual.conv1.weight": g("embeddings.patch_embedding.weight"), "visual.proj": hf_sd[prefix + "visual_projection.weight"].t().contiguous(),
          "visual.ln_pre.weight": g("pre_layrnorm.weight"), "visual.ln_pre.bias": g("pre_layrnorm.bias"),
          "visual.ln_post.weight": g("post_layernorm.weight"), "visual.ln_post.bias": g("post_layernorm.bias")}
    for i in range(layers):
        h, o = f"encoder.layers.{i}.", f"visual.transformer.resblocks.{i}."
        sd[o + "attn.in_proj_weight"] = torch.cat([g(h + f"self_attn.{n}_proj.weight") for n in "qkv"], 0)
        sd[o + "attn.in_proj_bias"] = torch.cat([g(h + f"self_attn.{n}_proj.bias") for n in "qkv"], 0)
        for a, b in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"),
                     ("mlp.c_proj", "mlp.fc2")):
            sd[o + a + ".weight"], sd[o + a + ".bias"] = g(h + b + ".weight"), g(h + b + ".bias")
    return sd


# CLIP preprocessing constants (modules.py:612-617); the resize itself (kornia, antialiased bicubic) stays with the caller
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
