"""tests/golden/wrapper_fullarch.pt: the reference's UNMODIFIED StreamingWrapper at the SHIPPED ARCHITECTURE (config.yaml:69-115: 4 levels,
channel_mult 1-2-4-4, attention at every level, 2 res blocks; 1.59 B + 0.68 B parameters) on a small latent (build container only).

    python oracle/make_golden_fullarch.py          # ~25 GB of RAM, a few minutes

The tiny golden (make_golden.py) has 2 levels and 1 res block; this run pins the oracle's 4-level wiring (skip connections, down / up
sampling, level-dependent attention, ControlNet + CAM at every resolution) against the real modules, and stores the reference outputs so
that tools/fullarch_parity.py can compare the HIP path with the REFERENCE at the shipped width.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle import svd_oracle as O  # noqa: E402
from oracle.cases import FULLARCH_CASE, full_unet_kwargs, fullarch_inputs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402

TOL = 5e-4


def load_by_name(module, seed):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    sd = init_by_name(s, seed=seed)
    module.load_state_dict(sd, strict=True)
    return sd


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    c = FULLARCH_CASE
    t0 = time.time()
    unet = VideoUNet(**full_unet_kwargs()).eval()
    sd_u = load_by_name(unet, seed=c["seed_unet"])
    cn = ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity", frame_expansion="none",
                              downsample_controlnet_cond=True, use_image_encoder_normalization=True, use_controlnet_mask=False,
                              condition_encoder="", conditioning_embedding_out_channels=[32, 96, 256, 512]).eval()
    sd_c = load_by_name(cn, seed=c["seed_cn"])
    print(f"reference modules built and loaded in {time.time() - t0:.0f} s "
          f"({sum(v.numel() for v in sd_u.values()) / 1e9:.2f} B + {sum(v.numel() for v in sd_c.values()) / 1e9:.2f} B parameters)")
    inp = fullarch_inputs()
    T, Tc = c["T"], c["Tc"]
    wrap = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=Tc)
    kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
    cond = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    t0 = time.time()
    ref = wrap(inp["x"], inp["t"], dict(cond), **dict(kw))
    xcat = torch.cat((inp["x"], inp["concat"]), 1)
    ref_nc = unet(xcat, inp["t"], context=inp["crossattn"], y=inp["vector"], num_video_frames=T, image_only_indicator=torch.zeros(2, T))
    print(f"reference forwards: {time.time() - t0:.0f} s")
    t0 = time.time()
    ora = O.streaming_wrapper(sd_u, sd_c, O.Cfg(), inp["x"], inp["t"], cond, 2, T, Tc, inp["ctrl_frames"])
    ora_nc = O.video_unet(sd_u, O.Cfg(), xcat, inp["t"], inp["crossattn"], inp["vector"], T)
    e, e_nc = (ref - ora).abs().max().item(), (ref_nc - ora_nc).abs().max().item()
    print(f"[full architecture] reference-vs-oracle max abs err: wrapper {e:.3e}, unet without control {e_nc:.3e} "
          f"(|ref| std {ref.std():.3f} / {ref_nc.std():.3f}); oracle {time.time() - t0:.0f} s")
    assert max(e, e_nc) <= TOL, (e, e_nc)
    path = os.path.join(ROOT, "tests", "golden", "wrapper_fullarch.pt")
    torch.save({"out": ref.clone(), "out_noctrl": ref_nc.clone()}, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
