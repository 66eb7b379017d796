"""The reference's OWN production precision at the SHIPPED ARCHITECTURE AND PROBLEM SIZE (build container only; ~30 GB RAM, tens of minutes):

    python oracle/measure_reference_autocast_fullsize.py      -> tests/golden/wrapper_fullsize_autocast.json

The unmodified reference StreamingWrapper (VideoUNet + ControlNet + CAM, 2.27 B parameters; CFG 2 x 25 frames @ 72x128 latent, ControlNet on
2 x 7 control frames of 576x1024) runs under torch.autocast("cpu", dtype=float16) -- its shipped `precision: 16-mixed` (config.yaml:8) --
and is compared with its own fp32 output, i.e. with the golden the GPU test uses (tests/golden/wrapper_fullsize.pt, written by
oracle/make_golden_fullsize.py from the same modules / seeds / inputs).  The result is the envelope row A5's tolerance is anchored to:
tests/test_gpu_fullsize_parity.py asserts that the HIP path is no further from the reference's fp32 output than this."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle.cases import FULLSIZE_CASE, full_unet_kwargs, fullsize_inputs  # noqa: E402
from oracle.make_golden_fullsize import load_by_name  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    c = FULLSIZE_CASE
    unet = VideoUNet(**full_unet_kwargs()).eval()
    load_by_name(unet, seed=c["seed_unet"])
    cn = ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity", frame_expansion="none",
                              downsample_controlnet_cond=True, use_image_encoder_normalization=True, use_controlnet_mask=False,
                              condition_encoder="", conditioning_embedding_out_channels=[32, 96, 256, 512]).eval()
    load_by_name(cn, seed=c["seed_cn"])
    inp = fullsize_inputs()
    T, Tc = c["T"], c["Tc"]
    wrap = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=Tc)
    kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
    cond = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "wrapper_fullsize.pt"))["out"]
    res = {"case": "StreamingWrapper.forward, CFG 2 x 25 frames @ 72x128 latent, ControlNet 2 x 7 frames @ 576x1024, shipped architecture",
           "reference": "unmodified reference modules on CPU; fp32 output = tests/golden/wrapper_fullsize.pt"}
    for name, dt in (("float16", torch.float16),):
        t0 = time.time()
        with torch.autocast("cpu", dtype=dt):
            out = wrap(inp["x"], inp["t"], dict(cond), **dict(kw)).float()
        e = (out - gold).flatten(1).pow(2).mean(1).sqrt()
        r = gold.flatten(1).pow(2).mean(1).sqrt()
        res[f"autocast_{name}"] = dict(l2_mean=e.mean().item(), l2_max=e.max().item(), rel_max=(e / r).max().item(), seconds=time.time() - t0)
        print(f"[reference StreamingWrapper, FULL SIZE, autocast {name} vs its own fp32] per-frame L2 abs mean {e.mean():.3e} max {e.max():.3e} | "
              f"rel max {(e / r).max():.3e} ({time.time() - t0:.0f} s)", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "wrapper_fullsize_autocast.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
