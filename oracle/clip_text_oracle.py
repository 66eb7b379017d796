"""fp32 restatement of transformers.CLIPTextModel(...).last_hidden_state (hidden_act "gelu"), the enhancer's text encoder
(code/i2v_enhance/pipeline_i2vgen_xl.py:250-347).  TEST INFRASTRUCTURE ONLY.  Pinned against the installed transformers package by
tests/test_oracle_golden.py::test_clip_text_oracle_matches_hf (the reference pins transformers==4.40.0; same architecture)."""
import torch
import torch.nn.functional as F


def text_tower(sd, input_ids, heads, prefix="text_model.", clip_skip=None):
    """clip_skip=k: hidden_states[-(k + 1)] (the last k encoder layers skipped) through final_layer_norm, as encode_prompt does for
    clip_skip is not None (pipeline_i2vgen_xl.py:246-260); the fork's __call__ default is clip_skip=1 (:645)."""
    g = lambda k: sd[prefix + k]
    B, L = input_ids.shape
    x = g("embeddings.token_embedding.weight")[input_ids] + g("embeddings.position_embedding.weight")[:L][None]
    W = x.shape[-1]
    mask = torch.triu(torch.full((L, L), float("-inf")), diagonal=1)
    n_layers = 0
    while f"{prefix}encoder.layers.{n_layers}.layer_norm1.weight" in sd:
        n_layers += 1
    i = 0
    while i < n_layers - (clip_skip or 0):
        b = f"encoder.layers.{i}."
        h = F.layer_norm(x, (W,), g(b + "layer_norm1.weight"), g(b + "layer_norm1.bias"), 1e-5)
        q, k, v = (F.linear(h, g(b + f"self_attn.{n}_proj.weight"), g(b + f"self_attn.{n}_proj.bias")) for n in "qkv")
        sp = lambda t: t.view(B, L, heads, W // heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=mask).transpose(1, 2).reshape(B, L, W)
        x = x + F.linear(o, g(b + "self_attn.out_proj.weight"), g(b + "self_attn.out_proj.bias"))
        h = F.layer_norm(x, (W,), g(b + "layer_norm2.weight"), g(b + "layer_norm2.bias"), 1e-5)
        x = x + F.linear(F.gelu(F.linear(h, g(b + "mlp.fc1.weight"), g(b + "mlp.fc1.bias"))), g(b + "mlp.fc2.weight"), g(b + "mlp.fc2.bias"))
        i += 1
    return F.layer_norm(x, (W,), g("final_layer_norm.weight"), g("final_layer_norm.bias"), 1e-5)
