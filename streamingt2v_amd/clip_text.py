"""CLIP text tower on the HIP kernels: ``prompt_embeds = text_encoder(input_ids)[0]`` of the enhancer
(code/i2v_enhance/pipeline_i2vgen_xl.py:250-347; transformers.CLIPTextModel, last_hidden_state).  state_dict keys are HuggingFace's
(``text_model.*``).  Pinned: transformers is installed in this image, so the oracle (oracle/clip_text_oracle.py) is checked against
the real CLIPTextModel on random weights (tests/test_oracle_golden.py).

77 tokens, head dim 64: per-head GEMM attention like clip_vision.py; the causal mask travels as the 16-bit RESIDUAL of the score GEMM
(0 / -30000: exp underflows to exactly 0), pad rows (77 -> 80) are masked as keys through the bias vector.
"""
import torch

from . import ops
from .params import Spec, check_state_dict
from .video_model import _dev_bf16, _dev_f32


class ClipTextConfig:
    def __init__(self, vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                 max_position_embeddings=77):
        self.vocab, self.width, self.mlp, self.layers, self.heads, self.npos = (vocab_size, hidden_size, intermediate_size, num_hidden_layers,
                                                                                 num_attention_heads, max_position_embeddings)
        self.hd = hidden_size // num_attention_heads
        assert self.hd % 32 == 0, "text tower head dim must be a multiple of 32 (64 for every CLIP text model in this pipeline)"


class CLIPTextTower:
    def __init__(self, cfg=None, prefix="text_model."):
        self.cfg, self.p = cfg or ClipTextConfig(), prefix

    def spec(self):
        c, p, s = self.cfg, self.p, Spec()
        s.add(p + "embeddings.token_embedding.weight", c.vocab, c.width); s.add(p + "embeddings.position_embedding.weight", c.npos, c.width)
        for i in range(c.layers):
            b = f"{p}encoder.layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                s.add(b + f"self_attn.{n}.weight", c.width, c.width); s.add(b + f"self_attn.{n}.bias", c.width)
            for n in ("layer_norm1", "layer_norm2"):
                s.add(b + n + ".weight", c.width); s.add(b + n + ".bias", c.width)
            s.add(b + "mlp.fc1.weight", c.mlp, c.width); s.add(b + "mlp.fc1.bias", c.mlp)
            s.add(b + "mlp.fc2.weight", c.width, c.mlp); s.add(b + "mlp.fc2.bias", c.width)
        s.add(p + "final_layer_norm.weight", c.width); s.add(p + "final_layer_norm.bias", c.width)
        return s

    def load_state_dict(self, sd, device="cuda", prefix=""):
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        sd = {k: v for k, v in sd.items() if k.startswith(self.p) and "position_ids" not in k}
        check_state_dict(self.spec(), sd)
        c, p, dev = self.cfg, self.p, device
        g = lambda k: sd[p + k].detach().float()
        self.tok, self.pos = g("embeddings.token_embedding.weight").to(dev), g("embeddings.position_embedding.weight").to(dev)
        self.ln_f = (_dev_f32(g("final_layer_norm.weight"), dev), _dev_f32(g("final_layer_norm.bias"), dev))
        self.blocks = []
        for i in range(c.layers):
            b = f"encoder.layers.{i}."
            W, Fv = (lambda k: _dev_bf16(g(b + k), dev)), (lambda k: _dev_f32(g(b + k), dev))
            self.blocks.append(dict(
                ln1=(Fv("layer_norm1.weight"), Fv("layer_norm1.bias")), ln2=(Fv("layer_norm2.weight"), Fv("layer_norm2.bias")),
                wqk=_dev_bf16(torch.cat([g(b + "self_attn.q_proj.weight"), g(b + "self_attn.k_proj.weight")], 0), dev),
                bqk=_dev_f32(torch.cat([g(b + "self_attn.q_proj.bias"), g(b + "self_attn.k_proj.bias")], 0), dev),
                wv=W("self_attn.v_proj.weight"), bv=Fv("self_attn.v_proj.bias"), wo=W("self_attn.out_proj.weight"), bo=Fv("self_attn.out_proj.bias"),
                w1=W("mlp.fc1.weight"), b1=Fv("mlp.fc1.bias"), w2=W("mlp.fc2.weight"), b2=Fv("mlp.fc2.bias")))
        self.device = dev
        return self

    def forward(self, input_ids, clip_skip=None):
        """input_ids int64 [B, L <= 77] -> last_hidden_state fp32 [B, L, width]; with clip_skip = k the last k encoder layers are skipped
        and the final LayerNorm is applied to hidden_states[-(k + 1)] (encode_prompt, pipeline_i2vgen_xl.py:246-260)."""
        c, dev = self.cfg, self.device
        B, L = input_ids.shape
        T = (L + 3) // 4 * 4
        tld = (T + 31) // 32 * 32
        H, hd, W = c.heads, c.hd, c.width
        emb = self.tok[input_ids.to(dev)] + self.pos[:L][None]                                   # gather: plumbing
        tok = torch.zeros((B, T, W), dtype=torch.float32, device=dev)
        tok[:, :L] = emb
        x = ops.to_elem(tok.view(B * T, W).contiguous())
        keymask = torch.zeros(T, dtype=torch.float32, device=dev)
        keymask[L:] = -1e30
        causal = torch.triu(torch.full((T, T), -30000.0, device=dev), diagonal=1).to(x.dtype).contiguous()
        vt = torch.zeros((B, W, tld), dtype=x.dtype, device=dev)
        s = torch.empty((T, T), dtype=torch.float32, device=dev)
        pm = torch.zeros((T, tld), dtype=x.dtype, device=dev)
        o = torch.empty((B * T, W), dtype=x.dtype, device=dev)
        for blk in self.blocks[: len(self.blocks) - (clip_skip or 0)]:
            n1 = ops.layernorm(x, *blk["ln1"])
            qk = ops.gemm(n1, blk["wqk"], bias=blk["bqk"])
            ops.gemm(n1, blk["wv"], bias=blk["bv"], trans_out=dict(tok_per_frame=T, tokens_ld=tld, out=vt))
            for b in range(B):
                rows = slice(b * T, (b + 1) * T)
                for h in range(H):
                    ops.gemm(qk[rows, h * hd:(h + 1) * hd], qk[rows, W + h * hd:W + (h + 1) * hd], bias=keymask, residual=causal, out=s)
                    ops.softmax_rows(s, pm[:, :T], hd ** -0.5)
                    ops.gemm(pm, vt[b, h * hd:(h + 1) * hd], out=o[rows, h * hd:(h + 1) * hd])
            x = ops.gemm(o, blk["wo"], bias=blk["bo"], residual=x)
            m = ops.gelu_(ops.gemm(ops.layernorm(x, *blk["ln2"]), blk["w1"], bias=blk["b1"]))
            x = ops.gemm(m, blk["w2"], bias=blk["b2"], residual=x)
        x = ops.layernorm(x, *self.ln_f)
        return x.float().view(B, T, W)[:, :L]

    __call__ = forward
