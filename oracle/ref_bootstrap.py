"""Import bootstrap for the *reference* StreamingSVD modules (oracle tooling, NOT product code).

TEST INFRASTRUCTURE ONLY.  Used solely by ``oracle/make_golden.py`` inside the build container to
run the unmodified reference (``/root/reference/code``) on CPU and dump golden vectors.  Nothing on
the GPU box imports this (``/root/reference`` does not exist there).

The reference imports packages that are absent here (pytorch_lightning, omegaconf, open_clip, kornia,
diffusers).  The stubs below satisfy the *imports only*; the one piece of third-party arithmetic the
hot path relies on -- ``diffusers.models.attention_processor.Attention`` (diffusers==0.30.2, pinned in
/root/reference/requirements.txt:6) as used by code/models/cam/conditioning.py:31-32,65-68 -- is
restated from its published behaviour (to_q/to_k/to_v without bias, to_out = [Linear(bias), Dropout],
heads = C/64, SDPA with scale d**-0.5).  That restatement is the only "parity unpinned" component.
"""
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference/code"


class _Attention(nn.Module):
    """diffusers==0.30.2 Attention + AttnProcessor2_0, restricted to what conditioning.py uses."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False,
                 upcast_attention=False, **kw):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv = cross_attention_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv, inner, bias=bias)
        self.to_v = nn.Linear(kv, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, encoder_hidden_states=None, attention_mask=None):
        c = x if encoder_hidden_states is None else encoder_hidden_states
        B, N, _ = x.shape
        h = self.heads
        q, k, v = (p(t).view(B, t.shape[1], h, -1).transpose(1, 2)
                   for p, t in ((self.to_q, x), (self.to_k, c), (self.to_v, c)))
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(B, N, -1)
        return self.to_out[1](self.to_out[0](o))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Make ``import models...`` of the reference work on CPU in this container."""
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "pytorch_lightning" not in sys.modules:
        _stub("pytorch_lightning", LightningModule=nn.Module, LightningDataModule=object)
    if "omegaconf" not in sys.modules:
        _stub("omegaconf", ListConfig=type("ListConfig", (list,), {}), OmegaConf=dict, DictConfig=dict)
    for n in ("open_clip", "kornia"):
        if n not in sys.modules:
            _stub(n)
    if "diffusers" not in sys.modules:
        _stub("diffusers")
        _stub("diffusers.models")
        _stub("diffusers.models.attention_processor", Attention=_Attention)
