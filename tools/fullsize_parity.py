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


CASES = ("sigma7.47", "s700", "s0p063")          # tests/golden/wrapper_fullsize.pt (round 2) + the two round-5 goldens at the ends of the AYS schedule


def _load_nets(sds, device):
    """the shipped-architecture VideoUNet + ControlNet with the by-name weights of FULLSIZE_CASE (CPU initialisation of 2.27 B parameters takes a
    minute: `sds` shares it between calls)"""
    from oracle.cases import FULLSIZE_CASE as c
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    cfg = UNetConfig()
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    if "u" not in sds:
        sds["u"], sds["c"] = init_by_name(unet.spec(), seed=c["seed_unet"]), init_by_name(cn.spec(), seed=c["seed_cn"])
    unet.load_state_dict(sds["u"], device=device)
    cn.load_state_dict(sds["c"], device=device)
    return unet, cn


def chunk_fullsize(dtype="fp16", device="cuda", sds=None):
    """Round 5: sampler o denoiser o guider o StreamingWrapper o VideoDecoder at the shipped size against tests/golden/chunk_fullsize.pt (the
    reference's own EulerEDMSampler / Denoiser / LinearPredictionGuider / wrapper / decoder: 2 AYS steps sigma 700 -> 0.002 -> 0, decode of the
    first 8 frames, clamp; oracle/make_golden_fullsize.py --which chunk).  Returns per-frame L2 of the decoded frames (on the golden's pixel
    subset) and of the latents z."""
    from oracle.cases import FULLSIZE_CASE as c, FULLSIZE_CHUNK_CASE as cc, fullsize_chunk_inputs, fullsize_pixel_subset
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VideoDecoder
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.set_grad_enabled(False)
    ops.set_element_dtype(DT[dtype])
    gold = torch.load(os.path.join(GOLD, "chunk_fullsize.pt"))
    sds = {} if sds is None else sds
    unet, cn = _load_nets(sds, device)
    dec = VideoDecoder()
    dec.load_state_dict(init_by_name(dec.spec(), seed=35), device=device)
    inp = fullsize_chunk_inputs()
    T, n = c["T"], cc["decode_frames"]
    dev = lambda d: {k: v.to(device) for k, v in d.items()}
    svd = StreamingSVD(StreamingWrapper(unet, cn, c["Tc"]), AutoencodingEngineDecoder(dec), sampler=EulerEDMSampler(num_steps=cc["steps"], num_frames=T))
    z = svd.sampler(svd.inference_model, inp["noise"].to(device).clone(), dev(inp["c"]), dev(inp["uc"]), batch_size=2, num_video_frames=T,
                    ctrl_frames=inp["ctrl_frames"].to(device))
    frames = svd.decode_first_stage(z[:n], clamp=True)
    torch.cuda.synchronize()
    idx = fullsize_pixel_subset(frames.shape[-2] * frames.shape[-1])
    rf = frame_errors(frames.float().flatten(2)[:, :, idx.to(device)], gold["frames_subset"])
    rz = frame_errors(z, gold["z"])
    del unet, cn, dec, svd
    torch.cuda.empty_cache()
    ops.set_element_dtype(None)
    return dict(frames=rf, z=rz)


def chunk30_fullsize(dtype="fp16", device="cuda", sds=None, also_chunk2=True):
    """Round 6: a WHOLE autoregressive chunk at the shipped size -- 30 AYS Euler steps of sampler o denoiser o guider o StreamingWrapper (ControlNet + 13 CAM
    mergers), decode_first_stage of all 25 frames, clamp (`_generate_conditional_output`, streaming_svd.py:155-221) -- and the chunk AFTER it, whose
    ctrl_frames are the last 7 frames THIS path decoded (`_autoregressive_generation`, :329-349), against tests/golden/chunk30_fullsize.pt and
    ar_handover_fullsize.pt (oracle/make_golden_fullsize_gpu.py).  also_chunk2: the round-5 two-step chunk against the reference's own CPU golden
    (tests/golden/chunk_fullsize.pt) on the same loaded networks.  Returns per-frame L2 dicts: chunk2 / chunk30 / handover, each {frames, z}."""
    from oracle.cases import (FULLSIZE_CASE as c, FULLSIZE_CHUNK30_CASE as c30, FULLSIZE_CHUNK_CASE as cc, fullsize_chunk30_inputs, fullsize_chunk_inputs,
                              fullsize_pixel_subset)
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VideoDecoder
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.set_grad_enabled(False)
    ops.set_element_dtype(DT[dtype])
    sds = {} if sds is None else sds
    unet, cn = _load_nets(sds, device)
    dec = VideoDecoder()
    dec.load_state_dict(init_by_name(dec.spec(), seed=35), device=device)
    T, Tc = c["T"], c["Tc"]
    dev = lambda d: {k: v.to(device) for k, v in d.items()}
    svd = StreamingSVD(StreamingWrapper(unet, cn, Tc), AutoencodingEngineDecoder(dec), sampler=EulerEDMSampler(num_steps=c30["steps"], num_frames=T))
    res = {}
    idx16 = fullsize_pixel_subset(576 * 1024)
    if also_chunk2:
        gold = torch.load(os.path.join(GOLD, "chunk_fullsize.pt"))
        inp = fullsize_chunk_inputs()
        z = svd.sampler(svd.inference_model, inp["noise"].to(device).clone(), dev(inp["c"]), dev(inp["uc"]), num_steps=cc["steps"], batch_size=2,
                        num_video_frames=T, ctrl_frames=inp["ctrl_frames"].to(device))
        fr = svd.decode_first_stage(z[:cc["decode_frames"]], clamp=True)
        res["chunk2"] = dict(frames=frame_errors(fr.float().flatten(2)[:, :, idx16.to(device)], gold["frames_subset"]), z=frame_errors(z, gold["z"]))
        del gold, fr, z
    idx = idx16[::2].contiguous().to(device)
    inp = fullsize_chunk30_inputs()
    cnd, uc = dev(inp["c"]), dev(inp["uc"])
    g1, g2 = torch.load(os.path.join(GOLD, "chunk30_fullsize.pt")), torch.load(os.path.join(GOLD, "ar_handover_fullsize.pt"))
    def chunk(ctrl, noise):          # the body of StreamingSVD._generate_conditional_output, keeping the latents for the report
        z = svd.sampler(svd.inference_model, noise.to(device).clone().float().contiguous(), cnd, uc, batch_size=2, num_video_frames=T, ctrl_frames=ctrl)
        return z, svd.decode_first_stage(z, clamp=True)
    z1, fr1 = chunk(inp["ctrl_frames"].to(device), inp["noise"])
    res["chunk30"] = dict(frames=frame_errors(fr1.float().flatten(2)[:, :, idx], g1["frames_subset"]), z=frame_errors(z1, g1["z"]))
    ctrl = StreamingSVD.extract_ctrl_frames(fr1, Tc)                      # the reference's hand-over, from THIS path's own frames
    res["handover_ctrl"] = frame_errors(ctrl[0].float().flatten(2)[:, :, idx], g2["ctrl_subset"])
    z2, fr2 = chunk(ctrl, inp["noise2"])
    torch.cuda.synchronize()
    res["handover"] = dict(frames=frame_errors(fr2.float().flatten(2)[:, :, idx], g2["frames_subset"]), z=frame_errors(z2, g2["z"]))
    del unet, cn, dec, svd, fr1, fr2
    torch.cuda.empty_cache()
    ops.set_element_dtype(None)
    return res


def wrapper_fullsize(dtype, device="cuda", sds=None, stream_f32=None, timing=False, plan=None, case="sigma7.47"):
    """stream_f32: None = the package default (ops.STREAM_F32), True / False = fp32 / 16-bit residual stream.  timing: also time the forward
    (3 runs after the parity run, device-synchronised) -> res['ms'].  case: which golden / input draw (CASES)."""
    from oracle.cases import FULLSIZE_CASE as c, fullsize_inputs, fullsize_inputs_sigma
    from streamingt2v_amd import ops
    torch.set_grad_enabled(False)
    ops.set_element_dtype(DT[dtype])
    prev_stream = ops.STREAM_F32
    if stream_f32 is not None:
        ops.set_stream_f32(stream_f32)
    prev_plan = (ops.EXACT_RIM, ops.CN_STREAM_F32, ops.STREAM_F32_MIN_CH)
    prev_kind = dict(ops.STREAM_F32_MIN_CH_KIND)
    prev_io = ops.STREAM_F32_SVT_IO_MIN_CH
    if plan is not None:              # (exact_rim, cn_stream_f32, stream_f32_min_ch[, per-kind thresholds]): the round-4 precision plan (None = the package default)
        ops.set_precision_plan(*plan[:3])
        ops.STREAM_F32_MIN_CH_KIND.update({"res": None, "svt": None})
        ops.STREAM_F32_SVT_IO_MIN_CH = 0
        if len(plan) > 3:
            kinds = dict(plan[3])
            ops.STREAM_F32_SVT_IO_MIN_CH = kinds.pop("svt_io", 0)
            ops.STREAM_F32_MIN_CH_KIND.update(kinds)
    try:
        return _wrapper_fullsize(dtype, device, {} if sds is None else sds, timing, case)
    finally:            # restore the process-wide plan on every path (an OOM in the forward must not leave a sweep's plan set for the next test)
        ops.set_stream_f32(prev_stream)
        ops.set_precision_plan(*prev_plan)
        ops.STREAM_F32_MIN_CH_KIND.update(prev_kind)
        ops.STREAM_F32_SVT_IO_MIN_CH = prev_io
        torch.cuda.empty_cache()


def _wrapper_fullsize(dtype, device, sds, timing, case):
    from oracle.cases import FULLSIZE_CASE as c, fullsize_inputs, fullsize_inputs_sigma
    from streamingt2v_amd import ops
    from streamingt2v_amd.wrappers import StreamingWrapper
    gold = torch.load(os.path.join(GOLD, "wrapper_fullsize.pt" if case == "sigma7.47" else f"wrapper_fullsize_{case}.pt"))
    unet, cn = _load_nets(sds, device)
    inp = {k: v.to(device) for k, v in (fullsize_inputs() if case == "sigma7.47" else fullsize_inputs_sigma(case)).items()}
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
    res["case"] = case
    del unet, cn, wrap
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
    ap.add_argument("--cases", default="sigma7.47", help="comma-separated subset of " + ",".join(CASES) + " or 'all'")
    ap.add_argument("--chunk", action="store_true", help="also the 2-step + decode chunk golden (tests/golden/chunk_fullsize.pt)")
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
            cases = CASES if a.cases == "all" else tuple(a.cases.split(","))
            for case, st, plan in [(cs, st, pl) for cs in cases for st in {"default": [None], "fp32": [True], "16": [False], "both": [True, False]}[a.stream] for pl in plans]:
                if not os.path.exists(os.path.join(GOLD, "wrapper_fullsize.pt" if case == "sigma7.47" else f"wrapper_fullsize_{case}.pt")):
                    continue
                r = wrapper_fullsize(name, sds=sds, stream_f32=st, timing=a.timing, plan=plan, case=case)
                print(f"[full-size StreamingWrapper.forward 2x25 @72x128 vs reference, case {case}, {name}, residual stream {'fp32' if r['stream_f32'] else '16 bit'}, "
                      f"exact rim {'on' if r['plan'][0] else 'off'}, ControlNet stream {'fp32' if r['plan'][1] else '16 bit'}, UNet fp32 stream at >= {r['plan'][2] or 'inf'} channels{(' ' + str(r['plan'][3])) if r['plan'][3] else ''}] "
                      f"per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} | rel max {r['rel_max']:.3e} | ref rms {r['ref_rms']:.3f} | "
                      f"corr {r['corr']:.7f}" + (f" | forward {r['ms']:.1f} ms" if "ms" in r else ""), flush=True)
        if a.chunk and os.path.exists(os.path.join(GOLD, "chunk_fullsize.pt")):
            r = chunk_fullsize(name, sds=sds)
            print(f"[full-size chunk: 2 Euler steps (sigma 700 -> 0.002 -> 0) + decode of 8 frames @576x1024 vs reference, {name}] decoded frames per-frame L2 abs max "
                  f"{r['frames']['abs_max']:.3e} mean {r['frames']['abs_mean']:.3e} (ref rms {r['frames']['ref_rms']:.3f}, corr {r['frames']['corr']:.7f}) | latents z abs max "
                  f"{r['z']['abs_max']:.3e} mean {r['z']['abs_mean']:.3e} (ref rms {r['z']['ref_rms']:.3f})", flush=True)


if __name__ == "__main__":
    main()
