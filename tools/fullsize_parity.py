"""Parity at the SHIPPED ARCHITECTURE AND PROBLEM SIZE against the REFERENCE's own outputs (tests/golden/*_fullsize.pt, written by
oracle/make_golden_fullsize.py from the unmodified reference modules on CPU):

  * StreamingWrapper.forward, CFG 2 x 25 frames @ 72x128 latent, ControlNet on 2 x 7 control frames of 576x1024   
  * VideoDecoder, 2 frames -> 576x1024 pixels (fp32 in the reference)                                              (row A11)

    python tools/fullsize_parity.py [--dtype bf16|fp16|both] [--which wrapper|vae|both]

Prints per-frame L2 (RMS error per frame, absolute on the network-output / [-1,1]-pixel scale, and relative).  The functions are also
what tests/test_gpu_fullsize_parity.py asserts on.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def frame_errors(out, ref):
    """per-frame RMS error, per-frame RMS of the reference, correlation."""
    out, ref = out.float().cpu(), ref.float().cpu()
    e = (out - ref).flatten(1).pow(2).mean(1).sqrt()
    r = ref.flatten(1).pow(2).mean(1).sqrt()
    corr = torch.corrcoef(torch.stack([out.flatten(), ref.flatten()]))[0, 1].item()
    return dict(abs_max=e.max().item(), abs_mean=e.mean().item(), rel_max=(e / r).max().item(), ref_rms=r.mean().item(), corr=corr)


def wrapper_fullsize(dtype, device="cuda", sds=None, stream_f32=None, timing=False, plan=None):
    """stream_f32: None = the package default (ops.STREAM_F32), True / False = fp32 / 16-bit residual stream.  timing: also time the forward
    (3 runs after the parity run, device-synchronised) -> res['ms']."""
    from oracle.cases import FULLSIZE_CASE as c, fullsize_inputs
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.set_grad_enabled(False)
    ops.set_element_dtype(DT[dtype])
    prev_stream = ops.STREAM_F32
    if stream_f32 is not None:
        ops.set_stream_f32(stream_f32)
    prev_plan = (ops.EXACT_RIM, ops.CN_STREAM_F32, ops.STREAM_F32_MIN_CH)
    prev_kind = dict(ops.STREAM_F32_MIN_CH_KIND)
    if plan is not None:              # (exact_rim, cn_stream_f32, stream_f32_min_ch[, per-kind thresholds]): the round-4 precision plan (None = the package default)
        ops.set_precision_plan(*plan[:3])
        ops.STREAM_F32_MIN_CH_KIND.update({"res": None, "svt": None})
        ops.STREAM_F32_SVT_IO_MIN_CH = 0
        if len(plan) > 3:
            kinds = dict(plan[3])
            ops.STREAM_F32_SVT_IO_MIN_CH = kinds.pop("svt_io", 0)
            ops.STREAM_F32_MIN_CH_KIND.update(kinds)
    gold = torch.load(os.path.join(GOLD, "wrapper_fullsize.pt"))
    cfg = UNetConfig()
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    if sds is None:
        sds = {}
    if "u" not in sds:             # by-name CPU initialisation of 2.27 B parameters takes a minute: shared between the element types
        sds["u"], sds["c"] = init_by_name(unet.spec(), seed=c["seed_unet"]), init_by_name(cn.spec(), seed=c["seed_cn"])
    unet.load_state_dict(sds["u"], device=device)
    cn.load_state_dict(sds["c"], device=device)
    inp = {k: v.to(device) for k, v in fullsize_inputs().items()}
    T = c["T"]
    wrap = StreamingWrapper(unet, cn, c["Tc"])
    out = wrap.forward(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")}, batch_size=2, num_video_frames=T,
                       image_only_indicator=torch.zeros(2, T, device=device), ctrl_frames=inp["ctrl_frames"])
    torch.cuda.synchronize()
    res = frame_errors(out, gold["out"])
    res["stream_f32"] = ops.STREAM_F32
    res["plan"] = (ops.EXACT_RIM, ops.CN_STREAM_F32, ops.STREAM_F32_MIN_CH, dict({k: v for k, v in ops.STREAM_F32_MIN_CH_KIND.items() if v is not None},
                                                                                 **({"svt_io": ops.STREAM_F32_SVT_IO_MIN_CH} if ops.STREAM_F32_SVT_IO_MIN_CH else {})))
    if timing:
        import time
        t0 = time.perf_counter()
        for _ in range(3):
            wrap.forward(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")}, batch_size=2, num_video_frames=T,
                         image_only_indicator=torch.zeros(2, T, device=device), ctrl_frames=inp["ctrl_frames"])
        torch.cuda.synchronize()
        res["ms"] = (time.perf_counter() - t0) / 3 * 1e3
    ops.set_stream_f32(prev_stream)
    ops.set_precision_plan(*prev_plan)
    ops.STREAM_F32_MIN_CH_KIND.update(prev_kind)
    ops.STREAM_F32_SVT_IO_MIN_CH = 0
    del unet, cn, wrap
    torch.cuda.empty_cache()
    return res


def decoder_fullsize(dtype, device="cuda"):
    from oracle.cases import fullsize_pixel_subset, fullsize_vae_inputs
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VideoDecoder
    torch.set_grad_enabled(False)
    ops.set_element_dtype(DT[dtype])
    gold = torch.load(os.path.join(GOLD, "vae_fullsize.pt"))
    dec = VideoDecoder()
    dec.load_state_dict(init_by_name(dec.spec(), seed=35), device=device)
    z = fullsize_vae_inputs()["z"].to(device)
    out = dec.forward(z, timesteps=z.shape[0])
    torch.cuda.synchronize()
    idx = fullsize_pixel_subset(out.shape[-2] * out.shape[-1])
    sub = out.float().flatten(2)[:, :, idx.to(device)]
    res = frame_errors(sub, gold["out_subset"])
    del dec
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="both")
    ap.add_argument("--which", default="both")
    ap.add_argument("--stream", default="default", choices=["default", "fp32", "16", "both"], help="residual stream of the wrapper: fp32 / 16 bit / both")
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--plans", default="default", help="'default' or 'sweep': the round-4 precision plan off / rim + ControlNet stream / + the UNet's fp32 stream at >= 1280, 640, 320 channels")
    a = ap.parse_args()
    plans = [None] if a.plans == "default" else [(False, False, 0), (True, True, 0), (True, True, 1280), (True, True, 640), (True, True, 320)]
    if a.plans == "level0":           # which KIND of level-0 block needs the fp32 stream: (rim, ControlNet stream, UNet >= 640) + ResBlocks / transformers at 320
        plans = [(True, True, 640, {"res": 320}), (True, True, 640, {"svt": 320}), (True, True, 320)]
    if a.plans == "level0io":         # level-0 ResBlocks in fp32 + the level-0 transformers' block output alone
        plans = [(True, True, 640, {"res": 320, "svt_io": 320}), (True, True, 320)]
    sds = {}
    for name in ("fp16", "bf16"):
        if a.dtype not in ("both", name):
            continue
        if a.which in ("both", "vae"):
            r = decoder_fullsize(name)
            print(f"[full-size VideoDecoder 2 frames @576x1024 vs reference, {name}] per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} | "
                  f"rel max {r['rel_max']:.3e} | ref rms {r['ref_rms']:.3f} | corr {r['corr']:.7f}", flush=True)
        if a.which in ("both", "wrapper"):
            for st, plan in [(st, pl) for st in {"default": [None], "fp32": [True], "16": [False], "both": [True, False]}[a.stream] for pl in plans]:
                r = wrapper_fullsize(name, sds=sds, stream_f32=st, timing=a.timing, plan=plan)
                print(f"[full-size StreamingWrapper.forward 2x25 @72x128 vs reference, {name}, residual stream {'fp32' if r['stream_f32'] else '16 bit'}, "
                      f"exact rim {'on' if r['plan'][0] else 'off'}, ControlNet stream {'fp32' if r['plan'][1] else '16 bit'}, UNet fp32 stream at >= {r['plan'][2] or 'inf'} channels{(' ' + str(r['plan'][3])) if r['plan'][3] else ''}] "
                      f"per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} | rel max {r['rel_max']:.3e} | ref rms {r['ref_rms']:.3f} | "
                      f"corr {r['corr']:.7f}" + (f" | forward {r['ms']:.1f} ms" if "ms" in r else ""), flush=True)


if __name__ == "__main__":
    main()
