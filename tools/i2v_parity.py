"""The enhancer UNet against the REFERENCE's vendored I2VGenXLUNet outputs, per precision plan:

    python tools/i2v_parity.py [--plans sweep|default] [--fullres] [--timing]

  * tests/golden/i2v_fullarch.pt : shipped architecture (1.42 B parameters), CFG 2 x 4 frames on a 9 x 16 latent
  * tests/golden/i2v_fullres.pt  : the same network at the SHIPPED latent size 90 x 160 (N = 14 400 spatial attention), CFG 2 x 4 frames
    (oracle/make_golden_i2v_fullarch.py [--fullres]: the unmodified vendored module on CPU, fp32)
Prints the per-frame L2 (RMS error of a frame's 4 x h x w prediction, absolute; the prediction has unit scale) for every plan
(ops.I2V_EXACT_RIM, ops.I2V_STREAM_F32_MIN_CH); --timing adds the time of one 2 x 38-frame forward at 90 x 160 (a blending window's step).
The functions are what tests/test_gpu_fullsize_parity.py asserts on.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def _frames(x):
    return x.permute(0, 2, 1, 3, 4).reshape(-1, x.shape[1], *x.shape[3:])


def frame_errors(out, ref):
    out, ref = _frames(out.float().cpu()), _frames(ref.float())
    e = (out - ref).flatten(1).pow(2).mean(1).sqrt()
    r = ref.flatten(1).pow(2).mean(1).sqrt()
    return dict(abs_max=e.max().item(), abs_mean=e.mean().item(), rel_max=(e / r).max().item(), ref_rms=r.mean().item())


def enhancer_parity(plan=None, fullres=False, sds=None, timing=False, device="cuda", dtype=torch.float16):
    """plan: None = package default, else (exact_rim, stream_f32_min_ch).  Returns the error dict (+ 'ms' of a 2 x 38-frame forward with timing)."""
    from oracle.cases import I2V_FULLARCH_CASE as ci, i2v_fullarch_inputs, i2v_fullres_inputs
    from streamingt2v_amd import ops
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    torch.set_grad_enabled(False)
    ops.set_element_dtype(dtype)
    prev = (ops.I2V_EXACT_RIM, ops.I2V_STREAM_F32_MIN_CH)
    try:
        if plan is not None:
            ops.set_i2v_precision_plan(*plan)
        eu = I2VGenXLUNet(I2VConfig())
        if sds is None:
            sds = {}
        if "i2v" not in sds:
            sds["i2v"] = init_by_name(eu.spec(), seed=ci["seed"])
        eu.load_state_dict(sds["i2v"], device=device)
        ei = i2v_fullres_inputs() if fullres else i2v_fullarch_inputs()
        gold = torch.load(os.path.join(GOLD, "i2v_fullres.pt" if fullres else "i2v_fullarch.pt"))
        out = eu(ei["sample"], ei["t"], fps=ei["fps"], image_latents=ei["image_latents"], image_embeddings=ei["image_embeddings"], encoder_hidden_states=ei["text"])[0]
        torch.cuda.synchronize()
        res = frame_errors(out, gold["out"])
        res["plan"] = (ops.I2V_EXACT_RIM, ops.I2V_STREAM_F32_MIN_CH)
        if timing:
            g = torch.Generator().manual_seed(1)
            Fr = 38
            eu.set_conditioning(ei["fps"], torch.randn(2, 4, Fr, 90, 160, generator=g) * 0.7, ei["image_embeddings"], ei["text"])
            fr = torch.randn(2 * Fr, 4, 90, 160, generator=g).to(device)
            eu.forward_frames(fr, 481.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                eu.forward_frames(fr, 481.0)
            torch.cuda.synchronize()
            res["ms"] = (time.perf_counter() - t0) / 2 * 1e3
        del eu
        torch.cuda.empty_cache()
        return res
    finally:
        ops.set_i2v_precision_plan(*prev)
        ops.set_element_dtype(None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plans", default="sweep")
    ap.add_argument("--fullres", action="store_true")
    ap.add_argument("--timing", action="store_true")
    a = ap.parse_args()
    plans = [None] if a.plans == "default" else [(False, 0), (True, 0), (True, 1280), (True, 640), (True, 320), (False, 320)]
    sds = {}
    have_fullres = a.fullres and os.path.exists(os.path.join(GOLD, "i2v_fullres.pt"))
    for pl in plans:
        for fullres in ([False, True] if have_fullres else [False]):
            r = enhancer_parity(pl, fullres=fullres, sds=sds, timing=a.timing and fullres == have_fullres)
            print(f"[I2VGenXLUNet.forward 2x4 frames @ {'90x160' if fullres else '9x16'} vs vendored reference, fp16, exact rim {'on' if r['plan'][0] else 'off'}, "
                  f"fp32 stream at >= {r['plan'][1] or 'inf'} channels] per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} | rel max {r['rel_max']:.3e} | "
                  f"ref rms {r['ref_rms']:.3f}" + (f" | 2x38-frame forward @90x160 {r['ms']:.0f} ms" if "ms" in r else ""), flush=True)


if __name__ == "__main__":
    main()
