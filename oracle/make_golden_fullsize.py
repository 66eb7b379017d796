"""FULL-SIZE goldens from the reference's UNMODIFIED modules (build container only; ~30 GB of RAM, ~10 minutes on 8 cores):

    python oracle/make_golden_fullsize.py [--which wrapper|vae|both|sigmas|chunk|round5]

  * StreamingWrapper.forward (code/models/diffusion/wrappers.py:23-78) at the shipped architecture AND the shipped problem size:
    CFG batch 2 x 25 frames, latent 72x128 (576x1024 pixels), ControlNet on 2 x 7 frames of 576x1024 control pixels
                                                                                  -> tests/golden/wrapper_fullsize.pt (1.8 MB)
  * VideoDecoder (code/models/svd/sgm/modules/autoencoding/temporal_ae.py:291-347) on a 2-frame 72x128 latent -> 2 x 3 x 576 x 1024
    pixels, fp32 like the reference's decode (config.yaml:310).  The output (14 MB) is stored on a seeded random 1/16 subset of the
    pixel positions of every frame (cases.fullsize_pixel_subset): 36 864 positions per frame and channel
                                                                                  -> tests/golden/vae_fullsize.pt (0.9 MB)
  * round 5 (--which sigmas): the same forward at sigma = 700 and sigma = 0.063 on fresh draws -> wrapper_fullsize_s700.pt, _s0p063.pt
  * round 5 (--which chunk): 2 Euler steps of the reference's sampler / denoiser / guider around the wrapper + fp32 decode of 8 frames
                                                                                  -> tests/golden/chunk_fullsize.pt
The timing lines this script prints are the reference's own CPU numbers (core count stated): profiles/r02_cpu_reference_forward.txt.
Inputs are re-derived from seeds (oracle/cases.py); weights by name (streamingt2v_amd.params.init_by_name), seeds as in FULLARCH_CASE.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_bootstrap  # noqa: E402

ref_bootstrap.install()
from oracle.cases import (FULLSIZE_CASE, FULLSIZE_CHUNK_CASE, FULLSIZE_SIGMA_CASES, full_unet_kwargs, fullsize_chunk_inputs, fullsize_inputs,  # noqa: E402
                          fullsize_inputs_sigma, fullsize_pixel_subset, fullsize_vae_inputs)
from streamingt2v_amd.params import Spec, init_by_name  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_by_name(module, seed):
    s = Spec()
    for k, v in module.state_dict().items():
        s.add(k, *v.shape)
    sd = init_by_name(s, seed=seed)
    module.load_state_dict(sd, strict=True)
    return sd


_BUILT = {}


def build_wrapper():
    """the reference's VideoUNet + ControlNet.from_unet + StreamingWrapper at the shipped architecture, weights by name (built once per process)"""
    if "wrap" in _BUILT:
        return _BUILT["wrap"]
    from models.control.controlnet import ControlNet
    from models.diffusion.video_model import VideoUNet
    from models.diffusion.wrappers import StreamingWrapper
    from models.svd.sgm.modules.diffusionmodules.wrappers import OpenAIWrapper
    c = FULLSIZE_CASE
    t0 = time.time()
    unet = VideoUNet(**full_unet_kwargs()).eval()
    load_by_name(unet, seed=c["seed_unet"])
    cn = ControlNet.from_unet(OpenAIWrapper(unet), merging_mode="addition", zero_conv_mode="Identity", frame_expansion="none",
                              downsample_controlnet_cond=True, use_image_encoder_normalization=True, use_controlnet_mask=False,
                              condition_encoder="", conditioning_embedding_out_channels=[32, 96, 256, 512]).eval()
    load_by_name(cn, seed=c["seed_cn"])
    print(f"reference modules built and loaded in {time.time() - t0:.0f} s", flush=True)
    _BUILT["wrap"] = StreamingWrapper(diffusion_model=unet, controlnet=cn, num_frame_conditioning=c["Tc"])
    return _BUILT["wrap"]


def build_decoder():
    if "dec" in _BUILT:
        return _BUILT["dec"]
    from models.svd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    kw = dict(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256,
              z_channels=4, double_z=True, attn_type="vanilla")
    dec = VideoDecoder(video_kernel_size=[3, 1, 1], **kw).eval()
    load_by_name(dec, seed=35)
    _BUILT["dec"] = dec
    return dec


def _forward(inp, tag):
    c = FULLSIZE_CASE
    T = c["T"]
    wrap = build_wrapper()
    kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
    cond = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    t0 = time.time()
    ref = wrap(inp["x"], inp["t"], dict(cond), **dict(kw))
    dt = time.time() - t0
    print(f"[cpu reference] StreamingWrapper.forward ({tag}), CFG 2 x {T} frames @ {c['h']}x{c['w']} latent, fp32, {torch.get_num_threads()} threads "
          f"on {os.cpu_count()} cores: {dt:.1f} s  (181.96 TFLOP algorithmic => {181.96 / dt:.3f} TFLOP/s); |out| std {ref.std():.4f}", flush=True)
    assert torch.isfinite(ref).all()
    return ref, dt


def wrapper():
    ref, dt = _forward(fullsize_inputs(), "sigma 7.47")
    path = os.path.join(OUT, "wrapper_fullsize.pt")
    torch.save({"out": ref.clone(), "cpu_seconds": dt, "threads": torch.get_num_threads()}, path)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def sigmas():
    """round 5: the same forward at both ends of the AYS schedule on fresh draws -> tests/golden/wrapper_fullsize_<name>.pt"""
    for name, cs in FULLSIZE_SIGMA_CASES.items():
        ref, dt = _forward(fullsize_inputs_sigma(name), f"sigma {cs['sigma']:g}")
        path = os.path.join(OUT, f"wrapper_fullsize_{name}.pt")
        torch.save({"out": ref.clone(), "sigma": cs["sigma"], "cpu_seconds": dt, "threads": torch.get_num_threads()}, path)
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def chunk():
    """round 5: sampler o denoiser o guider o wrapper o decoder at the shipped size -- the arithmetic of `_generate_conditional_output`
    (diffusion_trainer/streaming_svd.py:203-221) and `decode_first_stage` (:123-151) with the reference's own EulerEDMSampler
    (AlignYourSteps, 2 steps: sigma 700 -> 0.002 -> 0), Denoiser + VScalingWithEDMcNoise, LinearPredictionGuider (1.5 -> 3.0), on seeded
    conditioning; the first decode group (8 frames) is decoded in fp32 and clamped -> tests/golden/chunk_fullsize.pt"""
    from models.svd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from models.svd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    c, cc = FULLSIZE_CASE, FULLSIZE_CHUNK_CASE
    T = c["T"]
    wrap = build_wrapper()
    dec = build_decoder()
    inp = fullsize_chunk_inputs()
    den = Denoiser({"target": "models.svd.sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, num_steps=cc["steps"], verbose=False, device="cpu",
                              discretization_config={"target": "models.diffusion.discretizer.AlignYourSteps", "params": {"sigma_max": 700.0}},
                              guider_config={"target": "models.svd.sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                                             "params": {"max_scale": 3.0, "min_scale": 1.5, "num_frames": T}})
    add = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
    net_out = []

    def net(x, t, cd, **kw):
        o = wrap(x, t, cd, **kw)
        net_out.append(o.clone())
        return o
    t0 = time.time()
    z = sampler(lambda x, s, cd: den(net, x, s, cd, **dict(add)), inp["noise"].clone(), cond=inp["c"], uc=inp["uc"])
    t1 = time.time()
    n = cc["decode_frames"]
    frames = dec(z[:n].float() / 0.18215, timesteps=n).clamp(-1.0, 1.0)          # decode_first_stage: one group of 8, fp32 (config.yaml:310); clamp :221
    t2 = time.time()
    print(f"[cpu reference] {cc['steps']} Euler steps (2 x {T} frames each) {t1 - t0:.0f} s + decode of {n} frames {t2 - t1:.0f} s; |z| std {z.std():.4f} "
          f"|frames| std {frames.std():.4f}, clamped {100 * (frames.abs() == 1).float().mean():.2f} %", flush=True)
    idx = fullsize_pixel_subset(frames.shape[-2] * frames.shape[-1])
    path = os.path.join(OUT, "chunk_fullsize.pt")
    torch.save({"z": z.clone(), "frames_subset": frames.flatten(2)[:, :, idx].clone(), "frame_rms": frames.flatten(1).pow(2).mean(1).sqrt(),
                "net_out_step0": net_out[0].half(), "cpu_seconds": t2 - t0, "threads": torch.get_num_threads()}, path)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def chunk_autocast():
    """round 5: the SAME chunk with the network evaluations under torch.autocast("cpu", float16) -- the reference's shipped `precision: 16-mixed`
    (config.yaml:8); sampler state and decode in fp32 as the reference runs them (config.yaml:310) -- compared with its own fp32 chunk
    (tests/golden/chunk_fullsize.pt): the envelope the full-size chunk test is anchored to -> tests/golden/chunk_fullsize_autocast.json
    COST: fp16 autocast on CPU is far slower than the 807 s per forward measured for a single evaluation in round 2 -- in round 5 two runs in the 8-core
    build container were stopped after 3 h and 5.3 h (680 CPU-minutes) without reaching the end; the file is therefore not committed and the GPU test
    falls back to its literal bounds (tests/test_gpu_fullsize_parity.py::test_chunk_full_size_vs_reference)."""
    import json
    from models.svd.sgm.modules.diffusionmodules.denoiser import Denoiser
    from models.svd.sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    c, cc = FULLSIZE_CASE, FULLSIZE_CHUNK_CASE
    T = c["T"]
    wrap = build_wrapper()
    dec = build_decoder()
    inp = fullsize_chunk_inputs()
    gold = torch.load(os.path.join(OUT, "chunk_fullsize.pt"))
    den = Denoiser({"target": "models.svd.sgm.modules.diffusionmodules.denoiser_scaling.VScalingWithEDMcNoise"})
    sampler = EulerEDMSampler(s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, num_steps=cc["steps"], verbose=False, device="cpu",
                              discretization_config={"target": "models.diffusion.discretizer.AlignYourSteps", "params": {"sigma_max": 700.0}},
                              guider_config={"target": "models.svd.sgm.modules.diffusionmodules.guiders.LinearPredictionGuider",
                                             "params": {"max_scale": 3.0, "min_scale": 1.5, "num_frames": T}})
    add = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])

    def net(x, t, cd, **kw):
        with torch.autocast("cpu", dtype=torch.float16):
            return wrap(x, t, cd, **kw).float()
    t0 = time.time()
    z = sampler(lambda x, s, cd: den(net, x, s, cd, **dict(add)), inp["noise"].clone(), cond=inp["c"], uc=inp["uc"])
    n = cc["decode_frames"]
    frames = dec(z[:n].float() / 0.18215, timesteps=n).clamp(-1.0, 1.0)
    idx = fullsize_pixel_subset(frames.shape[-2] * frames.shape[-1])
    l2 = lambda a, b: (a.float() - b.float()).flatten(1).pow(2).mean(1).sqrt()
    ef, ez = l2(frames.flatten(2)[:, :, idx], gold["frames_subset"]), l2(z, gold["z"])
    res = {"case": "2 AYS Euler steps (sigma 700 -> 0.002 -> 0), CFG 2 x 25 frames @ 72x128, ControlNet on 2 x 7 frames @ 576x1024, decode of 8 frames; shipped architecture",
           "reference": "unmodified reference sampler / denoiser / guider / wrapper / decoder on CPU; fp32 run = tests/golden/chunk_fullsize.pt",
           "autocast_float16": dict(frames_l2_mean=ef.mean().item(), frames_l2_max=ef.max().item(), z_l2_mean=ez.mean().item(), z_l2_max=ez.max().item(),
                                    seconds=time.time() - t0)}
    print(f"[reference chunk, FULL SIZE, network under autocast float16 vs its own fp32] decoded frames per-frame L2 mean {ef.mean():.3e} max {ef.max():.3e} | "
          f"latents z mean {ez.mean():.3e} max {ez.max():.3e} ({time.time() - t0:.0f} s)", flush=True)
    path = os.path.join(OUT, "chunk_fullsize_autocast.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)


def vae():
    from models.svd.sgm.modules.autoencoding.temporal_ae import VideoDecoder
    kw = dict(ch=128, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=256,
              z_channels=4, double_z=True, attn_type="vanilla")
    dec = VideoDecoder(video_kernel_size=[3, 1, 1], **kw).eval()
    load_by_name(dec, seed=35)
    z = fullsize_vae_inputs()["z"]
    t0 = time.time()
    ref = dec(z, timesteps=z.shape[0])
    dt = time.time() - t0
    print(f"[cpu reference] VideoDecoder, {z.shape[0]} frames @ 576x1024, fp32, {torch.get_num_threads()} threads on {os.cpu_count()} cores: "
          f"{dt:.1f} s ({dt / z.shape[0]:.1f} s/frame; 6.94 TFLOP/frame => {6.94 * z.shape[0] / dt:.3f} TFLOP/s); |out| std {ref.std():.4f}", flush=True)
    idx = fullsize_pixel_subset(ref.shape[-2] * ref.shape[-1])
    sub = ref.flatten(2)[:, :, idx].clone()
    rms = ref.flatten(1).pow(2).mean(1).sqrt()
    path = os.path.join(OUT, "vae_fullsize.pt")
    torch.save({"out_subset": sub, "frame_rms": rms, "cpu_seconds": dt, "threads": torch.get_num_threads()}, path)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="both")
    a = ap.parse_args()
    torch.set_grad_enabled(False)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    if a.which in ("both", "vae"):
        vae()
    if a.which in ("both", "wrapper"):
        wrapper()
    if a.which in ("sigmas", "round5"):
        sigmas()
    if a.which in ("chunk", "round5"):
        chunk()
    if a.which == "chunk_autocast":
        chunk_autocast()


if __name__ == "__main__":
    main()
