"""Cross-check of oracle/clip_oracle.py against an INDEPENDENT implementation of the same architecture: HuggingFace transformers'
CLIPVisionModelWithProjection (installed in the build container; the reference pins transformers==4.40.0 for the enhancer's CLIP towers,
requirements.txt).  open_clip itself -- the package the reference's stage-1 conditioner imports -- is not available, so this is not a
pin against the reference's dependency; it shows that the restated ViT (class token, ln_pre, pre-LN blocks, fused in_proj, class pooling,
ln_post, bias-free projection) is the standard CLIP vision transformer.   python oracle/check_clip_vs_hf.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.clip_oracle import vision_tower  # noqa: E402


def hf_to_openclip(hf_sd, layers):
    g = lambda k: hf_sd["vision_model." + k]
    sd = {"visual.class_embedding": g("embeddings.class_embedding"), "visual.positional_embedding": g("embeddings.position_embedding.weight"),
          "visual.conv1.weight": g("embeddings.patch_embedding.weight"), "visual.proj": hf_sd["visual_projection.weight"].t().contiguous(),
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


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=320, intermediate_size=1280, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14,
                           projection_dim=64, hidden_act="gelu", layer_norm_eps=1e-5)
    hf = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():
        for p in hf.parameters():
            p.normal_(0, 0.05)
        img = torch.randn(2, 3, 56, 56)
        ref = hf(pixel_values=img).image_embeds
        out = vision_tower(hf_to_openclip(hf.state_dict(), 2), img, 4, 14)
    e = (ref - out).abs().max().item()
    print(f"[clip oracle vs HF CLIPVisionModelWithProjection] max abs err {e:.3e} (|ref| max {ref.abs().max():.3f})")
    assert e <= 2e-5, e
    if "--full" in sys.argv:       # ViT-H/14 as shipped: 32 layers x 1280, 16 heads of 80, 257 tokens, projection 1024 (632 M parameters)
        cfg = CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14,
                               projection_dim=1024, hidden_act="gelu", layer_norm_eps=1e-5)
        hf = CLIPVisionModelWithProjection(cfg).eval()
        with torch.no_grad():
            for p in hf.parameters():
                p.normal_(0, 0.02)
            img = torch.randn(1, 3, 224, 224)
            ref = hf(pixel_values=img).image_embeds
            out = vision_tower(hf_to_openclip(hf.state_dict(), 32), img, 16, 14)
        e = (ref - out).abs().max().item()
        print(f"[clip oracle vs HF, ViT-H/14 size] max abs err {e:.3e} (|ref| max {ref.abs().max():.3f}, rel {e / ref.abs().max().item():.2e})")
        assert e <= 1e-4 * max(1.0, ref.abs().max().item()), e


if __name__ == "__main__":
    main()
