"""tests/golden/apm_svt_tiny.pt: the reference's UNMODIFIED SpatialVideoTransformer with use_apm=True (BasicTransformerBlockWithAPM in the
spatial blocks, 17-token time context in the temporal block; code/models/svd/sgm/modules/attention.py:596-620, video_attention.py:174-333)
on a 17-token context -- pins oracle/svd_oracle.py's APM path (build container only):

    python oracle/make_golden_apm.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle import svd_oracle as O  # noqa: E402
from oracle.cases import apm_inputs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    from models.svd.sgm.modules.video_attention import SpatialVideoTransformer
    c = apm_inputs()
    C, T = c["C"], c["T"]
    svt = SpatialVideoTransformer(C, C // 64, 64, depth=1, context_dim=1024, time_context_dim=None, dropout=0.0, use_linear=True, attn_mode="softmax",
                                  use_spatial_context=True, ff_in=True, merge_strategy="learned_with_images", merge_factor=0.5,
                                  disable_self_attn=False, max_time_embed_period=10000, use_apm=True).eval()
    spec = Spec()
    for k, v in svt.state_dict().items():
        spec.add(k, *v.shape)
    sd = init_by_name(spec, seed=c["seed"])
    svt.load_state_dict(sd, strict=True)
    ref = svt(c["x"], context=c["context"], timesteps=T, image_only_indicator=torch.zeros(c["x"].shape[0] // T, T))
    ora = O.spatial_video_transformer(sd, "", c["x"], c["context"], T)
    e = (ref - ora).abs().max().item()
    print(f"[SpatialVideoTransformer use_apm, 17-token context] reference-vs-oracle max abs err {e:.3e} (|ref| std {ref.std():.3f})")
    assert e <= 2e-4, e
    from streamingt2v_amd.video_model import SpatialVideoTransformer as Ours
    ours = Spec()
    Ours("", C, 1024, use_apm=True).spec(ours)
    assert dict(ours) == dict(spec), (set(dict(ours)) ^ set(dict(spec)))
    path = os.path.join(ROOT, "tests", "golden", "apm_svt_tiny.pt")
    torch.save({"out": ref.clone()}, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
