"""The executed drop-in on the HIP kernels (-m gpu): INTEGRATION.md section 1 with libsvdhip.so behind it.

tests/test_dropin_reference.py executes `dropin.install` under the reference's UNMODIFIED `_generate_conditional_output` / `EulerEDMSampler` /
`Denoiser` / `decode_first_stage` -- in the build container, on fp32 torch statements of the launchers, because `/root/reference` does not travel to
the GPU box.  What travels is the RECORD of that run (tests/golden/dropin_calls_tiny.pt, written by oracle/make_golden_dropin.py through forward
hooks on the reference's own objects): every call the reference's code made on the two hot-path objects,

    inference_model(input * c_in, c_noise, cond, batch_size=, num_video_frames=, image_only_indicator=, ctrl_frames=)   denoiser.py:36-38, wrappers.py:23-78
    first_stage_model.decoder(z, timesteps=n)                                                                        autoencoder.py:210-212

with the tensors it passed and the tensors it got back.  Here the same objects are the HIP mirrors installed by `dropin.install` (HipModule shells
around StreamingWrapper / VideoDecoder on libsvdhip.so), and
  (1) every recorded call is REPLAYED through them with the recorded positional / keyword structure: the outputs match the reference's;
  (2) the loop is CLOSED on the HIP side: the Euler / guidance update of sampling.py:116-130 + guiders.py:60-99 (restated below from the recorded
      sigmas and guidance scales; checked against the recorded inputs of the next call) is driven by OUR network outputs, the result is decoded by
      OUR decoder through the reference's calling convention, and the frames match tests/golden/dropin_tiny.pt (the all-reference frames).
"""
import os
import types

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _dev(v):
    if torch.is_tensor(v):
        return v.cuda()
    if isinstance(v, dict):
        return {k: _dev(x) for k, x in v.items()}
    return v


def _rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).pow(2).mean().sqrt() / b.float().cpu().pow(2).mean().sqrt()).item()


@pytest.fixture(scope="module")
def installed():
    from oracle.cases import TINY_UNET
    from streamingt2v_amd import dropin, ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    torch.set_grad_enabled(False)
    ops.set_element_dtype(torch.float16)
    rec = torch.load(os.path.join(GOLD, "dropin_calls_tiny.pt"))
    c = rec["case"]
    ucfg = UNetConfig(num_res_blocks=TINY_UNET["num_res_blocks"], attention_resolutions=TINY_UNET["attention_resolutions"],
                      channel_mult=TINY_UNET["channel_mult"], conditioning_embedding_out_channels=TINY_UNET["cond_embed"])
    vcfg = VaeConfig(c["vae_ch"], c["vae_ch_mult"], c["vae_res"])
    # the checkpoint the reference modules of oracle/dropin_case.py were loaded from: by-name seeded tensors under the reference's key names
    sd = {}
    sd.update({"model.diffusion_model." + k: v for k, v in init_by_name(VideoUNet(ucfg).spec(), seed=1).items()})
    sd.update({"controlnet." + k: v for k, v in init_by_name(ControlNet(ucfg).spec(), seed=2).items()})
    sd.update({"first_stage_model.decoder." + k: v for k, v in init_by_name(VideoDecoder(vcfg).spec(), seed=3).items()})

    class Shell(nn.Module):          # the registered-children structure of the reference's StreamingSVD that install() swaps into
        pass
    model = Shell()
    model.inference_model = nn.Identity()
    model.first_stage_model = Shell()
    model.first_stage_model.decoder = nn.Identity()
    model.inference_params = types.SimpleNamespace(num_conditional_frames=c["Tc"])
    dropin.install(model, device="cuda", unet_cfg=ucfg, vae_cfg=vcfg, state_dict=sd)
    assert type(model.inference_model).__name__ == "HipModule" and type(model.first_stage_model.decoder).__name__ == "VideoDecoderModule"
    yield model, rec
    ops.set_element_dtype(None)


def test_recorded_reference_calls_replayed_through_the_hip_mirrors(installed):
    model, rec = installed
    n_net = n_dec = 0
    for kind, args, kwargs, out in rec["calls"]:
        if kind == "network":
            got = model.inference_model(*[_dev(a) for a in args], **{k: _dev(v) for k, v in kwargs.items()})
            e = _rel(got, out)
            print(f"[drop-in on HIP] network call {n_net} (kwargs {sorted(kwargs)}): relative L2 vs the reference's own output {e:.3e}")
            assert got.shape == out.shape and got.dtype == torch.float32 and e < 4e-3, e
            n_net += 1
        elif kind == "decoder":
            got = model.first_stage_model.decoder(*[_dev(a) for a in args], **{k: _dev(v) for k, v in kwargs.items()})
            e = (got.float().cpu() - out).flatten(1).pow(2).mean(1).sqrt().max().item()
            print(f"[drop-in on HIP] decoder call (kwargs {sorted(kwargs)}): per-frame L2 vs the reference's own output {e:.3e}")
            assert got.shape == out.shape and e < 2e-3, e
            n_dec += 1
    assert n_net == rec["steps"] and n_dec >= 1


def test_closed_loop_on_hip_matches_the_all_reference_frames(installed):
    model, rec = installed
    frames_ref = torch.load(os.path.join(GOLD, "dropin_tiny.pt"))["frames"]
    calls = {k: [c for c in rec["calls"] if c[0] == k] for k in ("network", "decoder", "sigmas", "guider_scale")}
    sig = calls["sigmas"][0][1][0].double()                      # AlignYourSteps sigmas incl. the final 0 (float64, sampling.py:50)
    gscale = calls["guider_scale"][0][1][0].float()              # LinearPredictionGuider.scale [1, T] or [T] (guiders.py:72-76)
    net = calls["network"]
    T = rec["case"]["T"]
    scaling = lambda s: (1.0 / (s * s + 1.0), -s / (s * s + 1.0) ** 0.5, 1.0 / (s * s + 1.0) ** 0.5, 0.25 * torch.log(torch.tensor(s)).item())

    def euler(x, out, i):
        """sampling.py:116-130 + denoiser.py:23-39 + guiders.py:86-99 on the CFG-doubled batch: x [T, ...], out [2T, ...] (uncond | cond)."""
        s, s_next = float(sig[i].float()), float(sig[i + 1].float())
        c_skip, c_out, _, _ = scaling(s)
        den = out * c_out + torch.cat([x, x]) * c_skip
        du, dc = den[:T], den[T:]
        g = gscale.reshape(-1, 1, 1, 1).to(x.device)
        d = (x - (du + g * (dc - du))) / s
        return x + d * (s_next - s)

    # the restated update reproduces the reference's own trajectory from the recorded outputs (the inputs of call i+1 are x_{i+1} * c_in_{i+1})
    x = net[0][1][0][:T] / scaling(float(sig[0].float()))[2]
    xs_ref = [x]
    for i in range(len(net) - 1):
        x = euler(x, net[i][3], i)
        nxt = net[i + 1][1][0][:T] / scaling(float(sig[i + 1].float()))[2]
        assert _rel(x, nxt) < 1e-4, (i, _rel(x, nxt))
        xs_ref.append(x)
    # closed loop on the HIP mirrors, the network called exactly as the reference's Denoiser calls it
    x = xs_ref[0].cuda()
    cond, kw = _dev(net[0][1][2]), {k: _dev(v) for k, v in net[0][2].items()}
    for i in range(len(net)):
        s = float(sig[i].float())
        _, _, c_in, c_noise = scaling(s)
        t = torch.full((2 * T,), c_noise, device="cuda")
        assert torch.allclose(t.cpu(), net[i][1][1].reshape(-1).float(), rtol=1e-5, atol=1e-6)
        out = model.inference_model(torch.cat([x, x]) * c_in, t, cond, **kw)
        x = euler(x, out, i)
    zdec, dkw = calls["decoder"][0][1][0], calls["decoder"][0][2]
    z = x / 0.18215                                              # decode_first_stage (streaming_svd.py:131)
    assert _rel(euler(xs_ref[-1], net[-1][3], len(net) - 1) / 0.18215, zdec) < 1e-4          # the recorded decoder input IS the loop's result
    frames = model.first_stage_model.decoder(z, **dkw).float().clamp(-1, 1).cpu()
    e = (frames - frames_ref).flatten(1).pow(2).mean(1).sqrt()
    print(f"[drop-in on HIP, closed loop: {len(net)} Euler steps + decode + clamp] per-frame L2 vs the all-reference frames: max {e.max():.3e} mean {e.mean():.3e}")
    assert frames.shape == frames_ref.shape and e.max().item() < 8e-3, e
