"""tests/golden/svd_conditioning_tiny.pt from the reference's UNMODIFIED stage-1 conditioning path (build container only).

    python oracle/make_golden_conditioner.py

`StreamingSVD._generate_conditional_output` (code/diffusion_trainer/streaming_svd.py:155-221) is run as an unbound method with the
reference's own `get_batch_sgm`, `get_unique_embedder_keys_from_conditioner` and a REAL `GeneralConditioner` built from config.yaml:160-218
-- real FrozenOpenCLIPImagePredictionEmbedder / ConcatTimestepEmbedderND / VideoPredictionEmbedderWithEncoder wrappers, with the two networks
inside them (OpenCLIP image tower, AutoencoderKLModeOnly) replaced by linear stand-ins -- around a recording sampler.  It pins what reaches
the sampler: c / uc (crossattn, concat, vector: key order, repeat over the 25 frames, zeroed unconditional entries), the noise shape, and the
extra model inputs (image_only_indicator, num_video_frames, batch_size, num_conditional_frames, ctrl_frames), plus decode + clamp.
"""
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ar_bootstrap  # noqa: E402
from oracle.cases import TINY_SVD_COND, fake_clip_embed, fake_cond_encode, tiny_svd_cond_inputs  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    Ref = ar_bootstrap.install()
    import importlib
    M = importlib.import_module("models.svd.sgm.modules.encoders.modules")

    class FakeClip(M.AbstractEmbModel):
        def forward(self, img):
            return fake_clip_embed(img)

    class FakeEncoder(nn.Module):
        def encode(self, x):
            return fake_cond_encode(x)

    fakes = types.ModuleType("oracle_ref_fakes")
    fakes.FakeClip, fakes.FakeEncoder = FakeClip, FakeEncoder
    sys.modules["oracle_ref_fakes"] = fakes
    P = "models.svd.sgm.modules.encoders.modules."
    cond = M.GeneralConditioner([
        dict(is_trainable=False, input_key="cond_frames_without_noise", target=P + "FrozenOpenCLIPImagePredictionEmbedder",
             params=dict(n_cond_frames=1, n_copies=1, open_clip_embedding_config=dict(target="oracle_ref_fakes.FakeClip", params={}))),
        dict(input_key="fps_id", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="motion_bucket_id", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
        dict(input_key="cond_frames", is_trainable=False, target=P + "VideoPredictionEmbedderWithEncoder",
             params=dict(disable_encoder_autocast=True, n_cond_frames=1, n_copies=1, is_ae=True, encoder_config=dict(target="oracle_ref_fakes.FakeEncoder", params={}))),
        dict(input_key="cond_aug", is_trainable=False, target=P + "ConcatTimestepEmbedderND", params=dict(outdim=256)),
    ])
    c = TINY_SVD_COND
    rec = {}

    class Sampler:
        guider = types.SimpleNamespace(num_frames=c["T"])

        def __call__(self, denoiser, randn, cond=None, uc=None):
            rec.update(randn=randn.clone(), c={k: v.clone() for k, v in cond.items()}, uc={k: v.clone() for k, v in uc.items()})
            denoiser(randn, torch.ones(randn.shape[0]), cond)
            return randn * 0.1

    def fake_denoiser(model, x, sigma, cc, **extra):
        rec["extra"] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in extra.items()}
        return x

    bare = types.SimpleNamespace(sampler=Sampler(), conditioner=cond, device="cpu", use_memopt=False, denoiser=fake_denoiser, inference_model="MODEL",
                                 inference_params=types.SimpleNamespace(num_conditional_frames=c["Tc"]),
                                 decode_first_stage=lambda z: torch.nn.functional.interpolate(z[:, :3] * 30.0, scale_factor=8, mode="nearest"))
    bare.get_batch_sgm = types.MethodType(Ref.get_batch_sgm, bare)
    bare.get_unique_embedder_keys_from_conditioner = types.MethodType(Ref.get_unique_embedder_keys_from_conditioner, bare)
    inp = tiny_svd_cond_inputs()
    torch.manual_seed(c["seed"])                       # cond_frames noise (rand_like) and the sampler noise (randn) come from the global stream
    out = Ref._generate_conditional_output(bare, inp["frame"], bare.inference_params, ctrl_frames=inp["ctrl_frames"])
    assert out.shape == (c["T"], 3, c["H"], c["W"]) and out.min() >= -1 and out.max() <= 1
    for k in ("crossattn", "concat", "vector"):
        print(f"[svd conditioning] c[{k}] {tuple(rec['c'][k].shape)}  uc[{k}] {tuple(rec['uc'][k].shape)}  |uc| max {rec['uc'][k].abs().max():.3f}")
    print("[svd conditioning] extra model inputs:", {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else v) for k, v in rec["extra"].items()})
    path = os.path.join(ROOT, "tests", "golden", "svd_conditioning_tiny.pt")
    first = lambda d: {k: v[:2].clone() for k, v in d.items()}          # frames are copies of each other: keep two rows, assert the rest here
    for d in (rec["c"], rec["uc"]):
        for k, v in d.items():
            assert v.shape[0] == c["T"] and all(torch.equal(v[0], v[i]) for i in range(1, c["T"])), k
    torch.save(dict(c=first(rec["c"]), uc=first(rec["uc"]), randn_shape=tuple(rec["randn"].shape),
                    extra={k: v for k, v in rec["extra"].items() if k != "ctrl_frames"}, ctrl_equal=torch.equal(rec["extra"]["ctrl_frames"], inp["ctrl_frames"])), path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
