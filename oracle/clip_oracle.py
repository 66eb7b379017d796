"""fp32 CPU restatement of the OpenCLIP ViT image tower (open_clip==2.24.0 VisionTransformer.forward as configured for ViT-H-14:
class-token pooling, ln_post after pooling, projection) used by the reference's FrozenOpenCLIPImageEmbedder
(code/models/svd/sgm/modules/encoders/modules.py:574-732).  TEST INFRASTRUCTURE ONLY.

open_clip is neither vendored under /root/reference nor installed here, and the reference holds no vectors for it:
**parity unpinned** against open_clip; cross-checked (4.8e-8) against HuggingFace transformers' CLIPVisionModelWithProjection on
mapped weights (oracle/check_clip_vs_hf.py, tests/test_oracle_golden.py) -- restated from the published model definition (transformer.py: VisionTransformer, ResidualAttentionBlock with
nn.MultiheadAttention(batch_first=False), nn.GELU, LayerNorm eps 1e-5, no layer scale, no patch dropout at inference)."""
import torch
import torch.nn.functional as F


def vision_tower(sd, images, heads, patch, prefix="visual."):
    g = lambda k: sd[prefix + k]
    B = images.shape[0]
    x = F.conv2d(images, g("conv1.weight"), stride=patch)                     # [B, width, grid, grid], no bias
    W = x.shape[1]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([g("class_embedding").view(1, 1, W).expand(B, 1, W), x], 1) + g("positional_embedding")
    x = F.layer_norm(x, (W,), g("ln_pre.weight"), g("ln_pre.bias"), 1e-5)
    i = 0
    while f"{prefix}transformer.resblocks.{i}.ln_1.weight" in sd:
        b = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), g(b + "ln_1.weight"), g(b + "ln_1.bias"), 1e-5)
        q, k, v = F.linear(h, g(b + "attn.in_proj_weight"), g(b + "attn.in_proj_bias")).chunk(3, -1)
        sp = lambda t: t.view(B, -1, heads, W // heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, -1, W)
        x = x + F.linear(o, g(b + "attn.out_proj.weight"), g(b + "attn.out_proj.bias"))
        h = F.layer_norm(x, (W,), g(b + "ln_2.weight"), g(b + "ln_2.bias"), 1e-5)
        x = x + F.linear(F.gelu(F.linear(h, g(b + "mlp.c_fc.weight"), g(b + "mlp.c_fc.bias"))), g(b + "mlp.c_proj.weight"), g(b + "mlp.c_proj.bias"))
        i += 1
    pooled = F.layer_norm(x[:, 0], (W,), g("ln_post.weight"), g("ln_post.bias"), 1e-5)
    return pooled @ g("proj")
