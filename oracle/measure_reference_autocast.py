"""How far is the REFERENCE'S OWN production precision from its fp32 path?  (build container only)

    python oracle/measure_reference_autocast.py

The reference runs the AR UNet under Lightning "16-mixed" autocast (config.yaml:8): fp16 GEMMs / convolutions, fp32 GroupNorm32, fp16
residual stream.  This script runs the UNMODIFIED reference StreamingWrapper (tiny configuration of oracle/cases.py) on CPU twice -- plain
fp32 and under torch.autocast(float16) -- and prints the per-frame L2 between the two: the deviation from the fp32 path that the reference
itself ships.  Printed next to tests/golden/wrapper_tiny.pt (the fp32 output the GPU tests compare against)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle.cases import TINY_UNET, tiny_unet_kwargs, tiny_wrapper_inputs  # noqa: E402
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402


def load_by_name(module, seed):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    module.load_state_dict(init_by_name(s, seed=seed), strict=True)


def main():
    torch.set_grad_enabled(False)
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    unet = VideoUNet(**tiny_unet_kwargs()).eval()
    load_by_name(unet, 1)
    cn = ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity", frame_expansion="none",
                              downsample_controlnet_cond=True, use_image_encoder_normalization=True, use_controlnet_mask=False,
                              condition_encoder="", conditioning_embedding_out_channels=list(TINY_UNET["cond_embed"])).eval()
    load_by_name(cn, 2)
    wrap = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=TINY_UNET["Tc"])
    inp = tiny_wrapper_inputs()
    kw = dict(batch_size=2, num_video_frames=TINY_UNET["T"], image_only_indicator=torch.zeros(2, TINY_UNET["T"]), ctrl_frames=inp["ctrl_frames"])
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    ref = wrap(inp["x"], inp["t"], dict(c), **dict(kw))
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "wrapper_tiny.pt"))["out"]
    print(f"fp32 run vs committed golden: max abs {(ref - gold).abs().max():.2e}")
    for dt in (torch.float16, torch.bfloat16):
        with torch.autocast("cpu", dtype=dt):
            out = wrap(inp["x"], inp["t"], dict(c), **dict(kw)).float()
        e = (out - ref).flatten(1).pow(2).mean(1).sqrt()
        r = ref.flatten(1).pow(2).mean(1).sqrt()
        print(f"[reference StreamingWrapper, tiny, autocast {str(dt)[6:]} vs its own fp32] per-frame L2 abs max {e.max():.3e} mean {e.mean():.3e} | rel max {(e / r).max():.3e}")


if __name__ == "__main__":
    main()
