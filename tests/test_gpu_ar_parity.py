"""Rows A2 / N2 on the GPU (-m gpu): the autoregressive outer loop with REAL chunks -- chunk 0 (no ControlNet, Karras/EDM schedule,
guidance 1 -> 3), PIL-grid quantisation, then AR chunks whose ControlNet + CAM consume the DECODED last frames of the previous
chunk (diffusion_trainer/streaming_svd.py:293-356, 388-394) -- HIP path vs the CPU oracle's video on identical injected noise.

The conditioner (CLIP tower + VAE encoder; SURVEY N4, pinned elsewhere) is replaced on BOTH sides by the same linear stand-ins
(oracle/cases.py fake_*), applied to each side's OWN anchor frame, so that every hand-over of the loop (anchor = chunk0[6],
ctrl_frames = last Tc decoded frames through the reference's range conversion, result[Tc:] kept) feeds back into the numbers.

Tolerances (round 3): errors compound over chunks (the decoded frames of chunk k drive the ControlNet of chunk k+1), so the bounds are per
chunk -- and they are no longer hand-picked numbers: tests/golden/ar_autocast_envelope.pt (oracle/make_golden_autocast_envelope.py) holds, for
exactly this case, (a) the video the UNMODIFIED reference networks / sampler / denoiser produce in fp32 and (b) how far the reference's OWN
production precision (fp16 autocast of the network evaluations, config.yaml:8) lands from it: per-chunk per-frame L2 and uint8 level
statistics.  The fp16 HIP path must be at least as close to the reference's fp32 video as the reference's own 16-mixed execution is, with 2
Euler steps per chunk and with the shipped 25 / 30 steps.  bf16 (selectable, 8x coarser rounding) is held to 8x that envelope.
"""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu
STEPS = 2
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ar_autocast_envelope.pt")


@pytest.fixture(scope="module", params=[torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def world(request):
    from oracle import cases, svd_oracle as O
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    ops.set_element_dtype(request.param)
    tu = cases.TINY_UNET
    tv = dict(ch=32, ch_mult=(1, 2, 2, 2), num_res_blocks=1)      # 4 levels: the decoder must upsample 8x (16x16 latent -> 128x128 control frames)
    cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                     channel_mult=tu["channel_mult"], conditioning_embedding_out_channels=tu["cond_embed"])
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    sd_u, sd_c = init_by_name(unet.spec(), seed=1), init_by_name(cn.spec(), seed=2)
    unet.load_state_dict(sd_u, device="cuda")
    cn.load_state_dict(sd_c, device="cuda")
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    sd_d = init_by_name(dec.spec(), seed=3)
    dec.load_state_dict(sd_d, device="cuda")
    T, Tc = tu["T"], tu["Tc"]
    model = StreamingSVD(StreamingWrapper(unet, cn, Tc), AutoencodingEngineDecoder(dec), EulerEDMSampler(num_steps=STEPS, num_frames=T),
                         num_conditional_frames=Tc)
    ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                 channel_mult=tu["channel_mult"], cond_embed_channels=tu["cond_embed"])
    vcfg = O.VaeCfg(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])
    g = torch.Generator(); g.manual_seed(2718)
    noises = [torch.randn(T, 4, tu["h"], tu["w"], generator=g) for _ in range(3)]
    image = torch.rand(3, 8 * tu["h"], 8 * tu["w"], generator=g) * 2 - 1
    vector = (torch.randn(1, 768, generator=g) * 0.5).repeat(T, 1)

    def conditioner(frame):
        """(c, uc) from ONE frame [3, H, W] on the frame's device: linear stand-ins for the CLIP tower and the VAE encoder."""
        emb = cases.fake_clip_embed(frame[None])
        lat = cases.fake_cond_encode(frame[None])
        v = vector.to(frame.device)
        c = dict(crossattn=emb[:, None].repeat(T, 1, 1), concat=lat.repeat(T, 1, 1, 1), vector=v)
        uc = dict(crossattn=torch.zeros_like(c["crossattn"]), concat=torch.zeros_like(c["concat"]), vector=v.clone())
        return c, uc

    yield dict(model=model, O=O, sd_u=sd_u, sd_c=sd_c, sd_d=sd_d, ocfg=ocfg, vcfg=vcfg, T=T, Tc=Tc, noises=noises, image=image,
               conditioner=conditioner, name=str(request.param)[6:], is16=request.param == torch.float16)
    ops.set_element_dtype(None)


def _l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return (got - ref).flatten(1).pow(2).mean(1).sqrt()


def _oracle_chunk0(w):
    O, T = w["O"], w["T"]
    c, uc = w["conditioner"](w["image"])
    net = lambda a, cn_, cc: O.video_unet(w["sd_u"], w["ocfg"], torch.cat((a, cc["concat"]), 1), cn_, cc["crossattn"], cc["vector"], T)
    z = O.euler_edm_sample(net, w["noises"][0].clone(), c, uc, STEPS, T, min_scale=1.0, max_scale=3.0, sigmas=O.edm_sigmas(STEPS))
    return O.decode_first_stage(w["sd_d"], w["vcfg"], z).clamp(-1, 1)


def test_initial_chunk_vs_oracle(world):
    """Row N2: _generate_initial_chunk (same VideoUNet without ControlNet / CAM, EDM/Karras sigmas, guidance 1 -> 3, decode, clamp)
    against the oracle's loop on the same noise."""
    w = world
    c, uc = w["conditioner"](w["image"].cuda())
    got = w["model"]._generate_initial_chunk(c, uc, w["noises"][0].cuda(), num_steps=STEPS)
    with torch.no_grad():
        ref = _oracle_chunk0(w)
    e = _l2(got, ref)
    print(f"[chunk 0 ({STEPS} EDM steps + decode) vs oracle, {w['name']}] per-frame L2 abs max {e.max():.3e} mean {e.mean():.3e}")
    assert torch.isfinite(got).all() and got.shape == ref.shape
    env = torch.load(GOLD)["steps"][STEPS]["envelope"]                  # the reference's own fp16 autocast on this chunk (8x for bf16)
    assert e.max().item() <= (1.0 if w["is16"] else 8.0) * env["l2_max"][0], (e.max().item(), env["l2_max"][0])


def test_autoregressive_chunks_vs_oracle(world):
    """Row A2: chunk 0 -> PIL grid -> 2 AR chunks (ControlNet on the previous chunk's decoded frames, CAM, anchor = chunk0[6]) -> video."""
    from streamingt2v_amd.streaming_svd import StreamingSVD
    w = world
    O, T, Tc, model = w["O"], w["T"], w["Tc"], w["model"]
    c, uc = w["conditioner"](w["image"].cuda())
    first = StreamingSVD.quantize_like_pil(model._generate_initial_chunk(c, uc, w["noises"][0].cuda(), num_steps=STEPS))
    video = model._autoregressive_generation(first, w["conditioner"], 2, [n.cuda() for n in w["noises"][1:]], num_steps=STEPS)
    assert video.shape[0] == T + 2 * (T - Tc)
    u8 = StreamingSVD.to_uint8_video(video).cpu()
    with torch.no_grad():
        chunks = [StreamingSVD.quantize_like_pil(_oracle_chunk0(w))]
        anchor = chunks[0][6]
        for k in range(2):
            ctrl = StreamingSVD.extract_ctrl_frames(chunks[-1], Tc)
            cc, cu = w["conditioner"](anchor)
            net = lambda a, cn_, cd: O.streaming_wrapper(w["sd_u"], w["sd_c"], w["ocfg"], a, cn_, cd, 2, T, Tc, ctrl)
            z = O.euler_edm_sample(net, w["noises"][1 + k].clone(), cc, cu, STEPS, T)
            chunks.append(O.decode_first_stage(w["sd_d"], w["vcfg"], z).clamp(-1, 1)[Tc:])
        ref = torch.cat(chunks, 0)
    from oracle.range_oracle import frames_to_uint8
    ref_u8 = frames_to_uint8(ref)
    e = _l2(video, ref)
    bounds = [0, T, T + (T - Tc), T + 2 * (T - Tc)]
    per_chunk = [e[bounds[i]:bounds[i + 1]].max().item() for i in range(3)]
    lvl = (u8.int() - ref_u8.int()).abs()
    print(f"[AR video: chunk 0 + 2 AR chunks vs oracle, {w['name']}] per-frame L2 abs max per chunk {per_chunk[0]:.3e} {per_chunk[1]:.3e} {per_chunk[2]:.3e}"
          f" | uint8: {100.0 * (lvl > 1).float().mean():.3f} % of bytes differ by > 1 level, max {lvl.max().item()}")
    # chunk 0 differs only through the 1/255 grid (a rounding flip = 7.8e-3 on single pixels); AR chunks inherit it through the ControlNet.
    # Bound = what the reference's own fp16 autocast does on this case (committed measurement), 8x for bf16.
    env = torch.load(GOLD)["steps"][STEPS]["envelope"]
    k = 1.0 if w["is16"] else 8.0
    for got_e, t in zip(per_chunk, env["l2_max"]):
        assert got_e <= k * t, (per_chunk, env["l2_max"])
    if w["is16"]:
        assert (lvl > 1).float().mean().item() <= env["u8_frac_gt1"], ((lvl > 1).float().mean().item(), env["u8_frac_gt1"])


def _hip_video(w, steps):
    from streamingt2v_amd.streaming_svd import StreamingSVD
    model = w["model"]
    c, uc = w["conditioner"](w["image"].cuda())
    first = StreamingSVD.quantize_like_pil(model._generate_initial_chunk(c, uc, w["noises"][0].cuda(), num_steps=min(steps, 25)))
    return model._autoregressive_generation(first, w["conditioner"], 2, [n.cuda() for n in w["noises"][1:]], num_steps=steps)


@pytest.mark.parametrize("steps", [2, 30])
def test_video_vs_reference_fp32_within_the_reference_autocast_envelope(world, steps):
    """The HIP video (chunk 0 + 2 AR hand-overs) against the video of the UNMODIFIED reference networks / sampler / denoiser in fp32
    (golden), bounded by the deviation of the reference's own fp16-autocast execution from that same video -- with 2 steps per chunk and with
    the shipped step counts (25 Euler-EDM steps for chunk 0, 30 AlignYourSteps steps per AR chunk)."""
    from oracle.range_oracle import frames_to_uint8
    w = world
    if steps == 30 and not w["is16"]:
        pytest.skip("the 30-step case runs in the parity element type (fp16)")
    g = torch.load(GOLD)
    assert g["case"]["T"] == w["T"] and g["case"]["Tc"] == w["Tc"]
    ref, env = g["steps"][steps]["video_sub"], g["steps"][steps]["envelope"]
    video = _hip_video(w, steps)[:, :, ::2, ::2].float().cpu()
    assert video.shape == ref.shape
    T, Tc = w["T"], w["Tc"]
    e = _l2(video, ref)
    bounds = [0, T, T + (T - Tc), T + 2 * (T - Tc)]
    per_chunk = [e[bounds[i]:bounds[i + 1]].max().item() for i in range(3)]
    lvl = (frames_to_uint8(video).int() - frames_to_uint8(ref).int()).abs()
    frac = (lvl > 1).float().mean().item()
    print(f"[AR video vs REFERENCE fp32, {steps} steps, {w['name']}] per-frame L2 max per chunk {per_chunk[0]:.3e} {per_chunk[1]:.3e} {per_chunk[2]:.3e} "
          f"(reference's own fp16 autocast: {env['l2_max'][0]:.3e} {env['l2_max'][1]:.3e} {env['l2_max'][2]:.3e}) | uint8 > 1 level: "
          f"{100 * frac:.3f} % (reference autocast {100 * env['u8_frac_gt1']:.3f} %), max {lvl.max().item()} ({env['u8_max']})")
    k = 1.0 if w["is16"] else 8.0
    for got_e, t in zip(per_chunk, env["l2_max"]):
        assert got_e <= k * t, (per_chunk, env["l2_max"])
    if w["is16"]:
        # no slack (round 3 needed 1.1x here): under the default precision plan the HIP video has 0.49 % of its bytes off by > 1 level against the
        # 0.82 % of the reference's own autocast execution at 30 steps
        assert frac <= env["u8_frac_gt1"], (frac, env["u8_frac_gt1"])


GOLD_SEEDS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ar_autocast_envelope_seeds.pt")


@pytest.mark.parametrize("seed", [31415, 16180])
def test_video_vs_reference_fp32_envelope_other_input_seeds(world, seed):
    """The 30-step case of the test above on two more draws of (noise, input image, vector) (round-3 review: one seed, thin margins): the HIP video
    against the unmodified reference's fp32 video of THAT draw, bounded by the reference's own fp16-autocast deviation on THAT draw
    (oracle/make_golden_autocast_envelope.py --steps 30 --seeds 31415 16180), per chunk and on the uint8 level statistic -- no slack factor."""
    from oracle import cases
    from oracle.range_oracle import frames_to_uint8
    from streamingt2v_amd.streaming_svd import StreamingSVD
    w = world
    if not w["is16"]:
        pytest.skip("the 30-step cases run in the parity element type (fp16)")
    g = torch.load(GOLD_SEEDS)[seed]
    ref, env = g["steps"][30]["video_sub"], g["steps"][30]["envelope"]
    T, Tc, tu = w["T"], w["Tc"], cases.TINY_UNET
    gen = torch.Generator(); gen.manual_seed(seed)
    noises = [torch.randn(T, 4, tu["h"], tu["w"], generator=gen) for _ in range(3)]
    image = torch.rand(3, 8 * tu["h"], 8 * tu["w"], generator=gen) * 2 - 1
    vector = (torch.randn(1, 768, generator=gen) * 0.5).repeat(T, 1)

    def conditioner(frame):
        emb, lat = cases.fake_clip_embed(frame[None]), cases.fake_cond_encode(frame[None])
        v = vector.to(frame.device)
        c = dict(crossattn=emb[:, None].repeat(T, 1, 1), concat=lat.repeat(T, 1, 1, 1), vector=v)
        return c, dict(crossattn=torch.zeros_like(c["crossattn"]), concat=torch.zeros_like(c["concat"]), vector=v.clone())
    model = w["model"]
    c, uc = conditioner(image.cuda())
    first = StreamingSVD.quantize_like_pil(model._generate_initial_chunk(c, uc, noises[0].cuda(), num_steps=25))
    video = model._autoregressive_generation(first, conditioner, 2, [n.cuda() for n in noises[1:]], num_steps=30)[:, :, ::2, ::2].float().cpu()
    assert video.shape == ref.shape
    e = _l2(video, ref)
    bounds = [0, T, T + (T - Tc), T + 2 * (T - Tc)]
    per_chunk = [e[bounds[i]:bounds[i + 1]].max().item() for i in range(3)]
    frac = ((frames_to_uint8(video).int() - frames_to_uint8(ref).int()).abs() > 1).float().mean().item()
    print(f"[AR video vs REFERENCE fp32, input seed {seed}, 25 + 30 steps, {w['name']}] per-frame L2 max per chunk {per_chunk[0]:.3e} {per_chunk[1]:.3e} "
          f"{per_chunk[2]:.3e} (reference's own fp16 autocast: {env['l2_max'][0]:.3e} {env['l2_max'][1]:.3e} {env['l2_max'][2]:.3e}) | uint8 > 1 level: "
          f"{100 * frac:.3f} % (reference autocast {100 * env['u8_frac_gt1']:.3f} %)")
    for got_e, t in zip(per_chunk, env["l2_max"]):
        assert got_e <= t, (per_chunk, env["l2_max"])
    assert frac <= env["u8_frac_gt1"], (frac, env["u8_frac_gt1"])
