"""Hot-path parity (-m gpu): the HIP execution of the StreamingSVD networks vs the CPU oracle / committed golden
vectors (reference outputs) on identical seeded weights and inputs.

Tolerance statement.  north_star asks for per-frame L2 <= 1e-3 against the fp32 reference.  Every kernel stores
16-bit activations and accumulates in fp32; the suite runs for both element types of the C ABI:
  * bf16 (north_star's "MFMA bf16 tiles", the bench default): one rounding is 2^-9, the measured block errors equal
    sqrt(#chained tensors) * 2^-9 (rounding-noise limited), so O(1) outputs cannot reach 1e-3 absolute; asserted:
    per-frame RMS error / per-frame RMS of the reference <= 2e-2 (sampler 3e-2, chunk 5e-2), correlation >= 0.9995.
  * fp16 (what the reference's own "16-mixed" autocast computes in, config.yaml:8; same MFMA rate): 8x finer.
    asserted: relative <= 4e-3 (sampler 6e-3, chunk 1e-2) AND absolute per-frame L2 of StreamingWrapper.forward vs the
    reference's output <= 1e-3 (round 4, default precision plan: measured 0.65e-3 mean, 0.70e-3 max; rounds 2 / 3: 0.95e-3 / 1.04e-3).
The measured values are printed (and tabulated in DESIGN.md) so that the gap to 1e-3 is visible, not hidden.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REL_L2 = 2e-2


def per_frame_rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = (got - ref).flatten(1).pow(2).mean(1).sqrt()
    r = ref.flatten(1).pow(2).mean(1).sqrt()
    return e, e / r


def report(name, got, ref, rel_tol=None):
    rel_tol = REL_L2 if rel_tol is None else rel_tol * (REL_L2 / 2e-2)
    assert torch.isfinite(got).all(), f"{name}: non-finite"
    e, rel = per_frame_rel_l2(got, ref)
    corr = torch.corrcoef(torch.stack([got.float().cpu().flatten(), ref.float().cpu().flatten()]))[0, 1].item()
    print(f"[{name} {str(ELEM)[6:]}] per-frame L2 abs max {e.max():.3e} mean {e.mean():.3e} | rel max {rel.max():.3e} | corr {corr:.6f}")
    assert rel.max().item() <= rel_tol, f"{name}: per-frame relative L2 {rel.max():.3e} > {rel_tol}"
    assert corr >= 0.9995, f"{name}: correlation {corr}"
    return e.max().item(), rel.max().item()


ELEM = torch.bfloat16


@pytest.fixture(scope="module", params=[torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def elem(request):
    """Both element types of the C ABI: bf16 (default, north_star) and fp16 (the reference's autocast precision)."""
    global ELEM, REL_L2
    from streamingt2v_amd import ops
    ELEM = request.param
    REL_L2 = 2e-2 if request.param == torch.bfloat16 else 4e-3
    ops.set_element_dtype(request.param)
    yield request.param
    ops.set_element_dtype(None)


@pytest.fixture(scope="module")
def tiny(elem):
    from oracle import cases, svd_oracle as O
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    tu = cases.TINY_UNET
    cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                     channel_mult=tu["channel_mult"], conditioning_embedding_out_channels=tu["cond_embed"])
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    sd_u, sd_c = init_by_name(unet.spec(), seed=1), init_by_name(cn.spec(), seed=2)
    unet.load_state_dict(sd_u, device="cuda")
    cn.load_state_dict(sd_c, device="cuda")
    wrap = StreamingWrapper(unet, cn, tu["Tc"])
    ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                 channel_mult=tu["channel_mult"], cond_embed_channels=tu["cond_embed"])
    return dict(unet=unet, cn=cn, wrap=wrap, sd_u=sd_u, sd_c=sd_c, ocfg=ocfg, tu=tu, cases=cases, O=O)


def _cuda(d):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}


def test_video_res_block(tiny):
    """One VideoResBlock (input_blocks.1.0) in isolation vs oracle."""
    O, unet, sd = tiny["O"], tiny["unet"], tiny["sd_u"]
    from streamingt2v_amd import ops
    T, H, W, Fr = 8, 16, 16, 16
    g = torch.Generator(); g.manual_seed(5)
    x = torch.randn(Fr, 320, H, W, generator=g)
    emb = torch.randn(Fr, 1280, generator=g)
    blk = unet.input_blocks[1][0]
    ref = O.video_res_block(sd, "input_blocks.1.0.", x.to(ELEM).float(), emb, T)
    tok = x.permute(0, 2, 3, 1).reshape(Fr * H * W, 320).to(ELEM).cuda().contiguous()
    out = blk.forward(tok, ops.to_bf16(emb.cuda().contiguous(), silu=True), Fr, T, H, W)
    out = out.float().view(Fr, H, W, -1).permute(0, 3, 1, 2)
    report("VideoResBlock", out, ref)


def test_spatial_video_transformer(tiny):
    O, unet, sd = tiny["O"], tiny["unet"], tiny["sd_u"]
    from streamingt2v_amd import ops
    T, H, W, Fr = 8, 16, 16, 16
    g = torch.Generator(); g.manual_seed(6)
    x = torch.randn(Fr, 320, H, W, generator=g)
    ctx = torch.randn(Fr, 1, 1024, generator=g)
    svt = unet.input_blocks[1][1]
    ref = O.spatial_video_transformer(sd, "input_blocks.1.1.", x.to(ELEM).float(), ctx, T)
    tok = x.permute(0, 2, 3, 1).reshape(Fr * H * W, 320).to(ELEM).cuda().contiguous()
    c, tc = unet._contexts(ctx.cuda(), T)
    out = svt.forward(tok, c, tc, Fr, T, H, W).float().view(Fr, H, W, -1).permute(0, 3, 1, 2)
    report("SpatialVideoTransformer", out, ref)


def test_cam_conditional_model(tiny):
    O, unet, sd = tiny["O"], tiny["unet"], tiny["sd_u"]
    T, Tc, H, W, B = 8, 3, 16, 16, 2
    g = torch.Generator(); g.manual_seed(7)
    s = torch.randn(B * T, 320, H, W, generator=g).to(ELEM).float()
    c = torch.randn(B * Tc, 320, H, W, generator=g).to(ELEM).float()
    ref = O.conditional_model(sd, "cross_attention_merger_input_blocks.1.", s, c, T, Tc)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, 320).to(ELEM).cuda().contiguous()
    out = unet.cross_attention_merger_input_blocks[1].forward(tok(s), tok(c), B * T, T, Tc, H, W)
    report("CAM ConditionalModel", out.float().view(B * T, H, W, -1).permute(0, 3, 1, 2), ref)


def test_controlnet_cond_embedding(tiny):
    O, cn, sd = tiny["O"], tiny["cn"], tiny["sd_c"]
    g = torch.Generator(); g.manual_seed(8)
    cond = torch.rand(4, 3, 64, 64, generator=g) * 2 - 1
    ref = O.controlnet_cond_embedding(sd, tiny["ocfg"], cond)
    out, H, W = cn.controlnet_cond_embedding.forward(cond.cuda())
    report("ControlNet cond embedding", out.float().view(4, H, W, -1).permute(0, 3, 1, 2), ref)


def test_streaming_wrapper_vs_reference_golden(tiny, golden_dir):
    """StreamingWrapper.forward (ControlNet + UNet + CAM) against the REFERENCE's output (tests/golden)."""
    gold = torch.load(os.path.join(golden_dir, "wrapper_tiny.pt"))
    inp = _cuda(tiny["cases"].tiny_wrapper_inputs())
    tu = tiny["tu"]
    out = tiny["wrap"].forward(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")},
                               batch_size=2, num_video_frames=tu["T"],
                               image_only_indicator=torch.zeros(2, tu["T"], device="cuda"), ctrl_frames=inp["ctrl_frames"])
    e_abs, _ = report("StreamingWrapper.forward vs reference", out, gold["out"])
    if ELEM == torch.float16:
        # north_star: per-frame L2 <= 1e-3 vs the reference, asserted literally.  Measured 0.65e-3 mean / 0.70e-3 max in fp16 under the default
        # precision plan of round 4 (1.04e-3 max with the all-16-bit arithmetic of rounds 2 / 3).
        assert e_abs <= 1e-3, e_abs
    # no-ControlNet path (config C2): VideoUNet.forward with hs_control_* = None
    x = torch.cat((inp["x"], inp["concat"]), 1)
    out = tiny["unet"].forward(x, inp["t"], context=inp["crossattn"], y=inp["vector"], num_video_frames=tu["T"],
                               image_only_indicator=torch.zeros(2, tu["T"], device="cuda"))
    report("VideoUNet.forward (no control) vs reference", out, gold["out_noctrl"])


def test_sampler_vs_reference_golden(tiny, golden_dir):
    """2 Euler-EDM steps (AYS schedule, CFG 1.5->3.0) through the fused sampler vs the reference sampler stack."""
    from streamingt2v_amd.sampling import EulerEDMSampler
    gold = torch.load(os.path.join(golden_dir, "sampler_tiny.pt"))
    tu = tiny["tu"]
    sin = tiny["cases"].tiny_sampler_inputs()
    inp = _cuda(tiny["cases"].tiny_wrapper_inputs())
    sampler = EulerEDMSampler(num_steps=2, num_frames=tu["T"])
    z = sampler(tiny["wrap"], sin["noise"].cuda().clone(), _cuda(sin["c"]), _cuda(sin["uc"]), batch_size=2,
                num_video_frames=tu["T"], ctrl_frames=inp["ctrl_frames"])
    report("EulerEDMSampler 2 steps vs reference", z, gold["z"], rel_tol=3e-2)   # |x| ~ 700 at step 0 amplifies rounding


def test_vae_decoder_vs_reference_golden(elem, golden_dir):
    from oracle import cases
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VaeConfig, VideoDecoder
    tv = cases.TINY_VAE
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    dec.load_state_dict(init_by_name(dec.spec(), seed=3), device="cuda")
    gold = torch.load(os.path.join(golden_dir, "vae_tiny.pt"))
    z = cases.tiny_vae_inputs()["z"].cuda()
    out = dec.forward(z, timesteps=z.shape[0])
    report("VideoDecoder vs reference", out, gold["out"])


def test_config1_end_to_end_vs_oracle(tiny):
    """BASELINE config 1 shape of work on the tiny nets: one chunk = sampler steps + temporal VAE decode + clamp,
    HIP path vs the CPU oracle on identical noise (frames compared before uint8 quantisation)."""
    O, tu, cases = tiny["O"], tiny["tu"], tiny["cases"]
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VaeConfig, VideoDecoder
    tv = cases.TINY_VAE
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    sd_d = init_by_name(dec.spec(), seed=3)
    dec.load_state_dict(sd_d, device="cuda")
    T, steps = tu["T"], 4
    model = StreamingSVD(tiny["wrap"], AutoencodingEngineDecoder(dec), EulerEDMSampler(num_steps=steps, num_frames=T),
                         num_conditional_frames=tu["Tc"])
    sin = cases.tiny_sampler_inputs()
    inp = cases.tiny_wrapper_inputs()
    frames = model._generate_conditional_output(_cuda(sin["c"]), _cuda(sin["uc"]), inp["ctrl_frames"].cuda(),
                                                sin["noise"].cuda(), num_steps=steps)
    net = lambda a, cn_, cc: O.streaming_wrapper(tiny["sd_u"], tiny["sd_c"], tiny["ocfg"], a, cn_, cc, 2, T, tu["Tc"],
                                                 inp["ctrl_frames"])
    z = O.euler_edm_sample(net, sin["noise"].clone(), sin["c"], sin["uc"], steps, T)
    ref = O.decode_first_stage(sd_d, O.VaeCfg(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]), z).clamp(-1, 1)
    assert frames.shape == ref.shape and frames.shape[:2] == (T, 3)
    report("config-1 chunk (4 steps + decode) vs oracle", frames, ref, rel_tol=5e-2)


def test_cfg_pair_split_equals_full_batch(tiny):
    """The 2-rank CFG split (parallel.CfgPairExchange) on one device: a test double evaluates BOTH halves in turn
    through the half-batch path; the sampler result must equal the ordinary CFG-batch-2 sampler (same kernels, same
    accumulation order -> agreement to a few ulp of the fp32 state)."""
    from streamingt2v_amd.sampling import EulerEDMSampler
    tu, cases, wrap = tiny["tu"], tiny["cases"], tiny["wrap"]
    sin = cases.tiny_sampler_inputs()
    inp = _cuda(cases.tiny_wrapper_inputs())
    kw = dict(batch_size=2, num_video_frames=tu["T"], ctrl_frames=inp["ctrl_frames"])
    z_full = EulerEDMSampler(num_steps=2, num_frames=tu["T"])(wrap, sin["noise"].cuda().clone(), _cuda(sin["c"]), _cuda(sin["uc"]), **kw)

    class BothHalves:
        """Stands in for the partner rank: computes the other half locally instead of receiving it."""
        half = 1
        def __init__(self):
            self.other = None
        def gather(self, net_cond):
            return torch.cat([self.other(), net_cond], 0)

    ex = BothHalves()
    sampler = EulerEDMSampler(num_steps=2, num_frames=tu["T"], cfg_exchange=ex)
    x = sin["noise"].cuda().clone()
    uc, c = _cuda(sin["uc"]), _cuda(sin["c"])
    # re-implement the loop body's "partner" evaluation through the same half-batch entry point
    import numpy as np
    sig = sampler.sigmas()
    x.mul_(float(np.sqrt(1.0 + sig[0] ** 2.0)))
    from streamingt2v_amd import ops
    g = sampler.guider.scale.cuda().float().contiguous()
    for i in range(len(sig) - 1):
        s, s_next = float(np.float32(sig[i])), float(np.float32(sig[i + 1]))
        _, _, c_in, c_noise = sampler.scaling(s)
        halves = [wrap.forward_fused(x, c_in, c_noise, {k: h[k].float().contiguous() for k in ("vector", "crossattn", "concat")},
                                     batch_size=1, num_video_frames=tu["T"], ctrl_frames=inp["ctrl_frames"]) for h in (uc, c)]
        ops.edm_euler_step(x, torch.cat(halves, 0), g, s, s_next)
    # not bit-identical: M halves, so tuned tile shapes and GroupNorm chunking differ -> a different (equally valid) rounding path
    report("CFG-pair split vs CFG batch 2 (2 sampler steps)", x, z_full, rel_tol=3e-2)


def test_range_quantisation_bit_exact():
    """Row A13: [-1,1] frames -> uint8 NHWC, bit for bit the reference's convert_range + IImage/torch2np output (golden bytes
    produced by the unmodified reference functions, oracle/make_golden_range.py) and the CPU oracle."""
    from oracle import cases
    from oracle.range_oracle import frames_to_uint8
    from streamingt2v_amd.streaming_svd import StreamingSVD
    x = cases.range_inputs()
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "range_tiny.pt"))["u8"]
    got = StreamingSVD.to_uint8_video(x.cuda()).cpu()
    assert got.dtype == torch.uint8 and torch.equal(got, gold)
    g = torch.Generator(); g.manual_seed(7)
    big = (torch.rand(3, 3, 576, 1024, generator=g) * 2.2 - 1.1).clamp(-1, 1)        # full frame size incl. saturated values
    assert torch.equal(StreamingSVD.to_uint8_video(big.cuda()).cpu(), frames_to_uint8(big))


def test_vae_encoder_vs_reference_golden(elem, golden_dir):
    """The conditioner's VAE encoder (sgm Encoder, asymmetric-pad stride-2 convs) on the HIP kernels vs the reference output."""
    from oracle import cases
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import Encoder, VaeConfig
    tv = cases.TINY_VAE
    enc = Encoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    enc.load_state_dict(init_by_name(enc.spec(), seed=4), device="cuda")
    out = enc(cases.tiny_vae_inputs()["x_enc"].cuda())
    gold = torch.load(os.path.join(golden_dir, "vae_enc_tiny.pt"))["out"]
    report("VAE Encoder vs reference", out, gold)


def test_cond_frame_encoder_vs_reference_golden(elem, golden_dir):
    """AutoencoderKLModeOnly.encode of the conditioner's cond_frames embedder (encoder + quant_conv + posterior mode)."""
    from oracle import cases
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig
    tv = cases.TINY_VAE
    enc = CondFrameEncoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    enc.load_state_dict(init_by_name(enc.spec(), seed=6), device="cuda")
    out = enc(cases.tiny_vae_inputs()["x_enc"].cuda())
    report("cond-frame encoder vs reference", out, torch.load(os.path.join(golden_dir, "cond_enc_tiny.pt"))["out"])


def test_clip_vision_tower_vs_oracle(elem):
    """OpenCLIP ViT image tower (head dim 80 -> per-head GEMM attention with padded heads and masked pad tokens) vs its CPU
    restatement (open_clip is not vendored: parity unpinned, see oracle/clip_oracle.py); tiny: width 320, 4 heads of 80, 2 layers."""
    from oracle.clip_oracle import vision_tower
    from streamingt2v_amd.clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from streamingt2v_amd.params import init_by_name
    cfg = ClipVisionConfig(width=320, layers=2, heads=4, patch_size=14, image_size=56, embed_dim=64)
    tower = OpenCLIPVisionTower(cfg)
    sd = init_by_name(tower.spec(), seed=8)
    tower.load_state_dict(sd, device="cuda")
    g = torch.Generator(); g.manual_seed(3)
    img = torch.randn(2, 3, 56, 56, generator=g)
    out = tower(img.cuda())
    with torch.no_grad():
        ref = vision_tower(sd, img, cfg.heads, cfg.patch)
    assert out.shape == ref.shape == (2, 64)
    report("OpenCLIP vision tower", out[:, None], ref[:, None])


def test_native_conditioner_and_front_end_single_chunk(elem):
    """StreamingPipeline.image_to_video with the NATIVE conditioner (CLIP tower + cond-frame encoder + sinusoid vector) for one chunk:
    conditioning tensors vs the CPU oracles of their parts, and the decoded uint8 video has the reference's shape/dtype contract."""
    from oracle import cases, svd_oracle as O
    from oracle.clip_oracle import vision_tower
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from streamingt2v_amd.conditioner import SVDConditioner
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    tu, tv = cases.TINY_UNET, cases.TINY_VAE
    ucfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                      channel_mult=tu["channel_mult"], conditioning_embedding_out_channels=tu["cond_embed"])
    unet, cnet = VideoUNet(ucfg), ControlNet(ucfg)
    unet.load_state_dict(init_by_name(unet.spec(), seed=1), device="cuda")
    cnet.load_state_dict(init_by_name(cnet.spec(), seed=2), device="cuda")
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    dec.load_state_dict(init_by_name(dec.spec(), seed=3), device="cuda")
    ccfg, ecfg = ClipVisionConfig(width=320, layers=2, heads=4, image_size=224, embed_dim=1024), VaeConfig(32, (1, 1, 1, 2), 1)
    clip, enc = OpenCLIPVisionTower(ccfg), CondFrameEncoder(ecfg)
    sd_c, sd_e = init_by_name(clip.spec(), seed=8), init_by_name(enc.spec(), seed=9)
    clip.load_state_dict(sd_c, device="cuda"); enc.load_state_dict(sd_e, device="cuda")
    T = tu["T"]
    cond = SVDConditioner(clip, enc, num_frames=T, generator=torch.Generator(device="cuda").manual_seed(5))
    g = torch.Generator(); g.manual_seed(6)
    frame = torch.rand(3, 8 * tu["h"], 8 * tu["w"], generator=g) * 2 - 1
    c, uc = cond(frame.cuda())
    # parts vs their oracles (same uniform noise regenerated from the same device generator state)
    noise = torch.rand((1,) + tuple(frame.shape), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda").cpu()
    with torch.no_grad():
        ref_cross = vision_tower(sd_c, SVDConditioner.clip_preprocess(frame[None]), ccfg.heads, ccfg.patch)
        ref_cat = O.cond_frame_encode(sd_e, O.VaeCfg(32, (1, 1, 1, 2), 1), frame[None] + 0.02 * noise)
    report("conditioner crossattn (CLIP)", c["crossattn"][:1], ref_cross[:, None])
    report("conditioner concat (VAE mode)", c["concat"][:1], ref_cat)
    assert c["concat"].shape == (T, 4, tu["h"], tu["w"]) and uc["concat"].abs().sum().item() == 0
    pipe = P.StreamingPipeline(unet, cnet, dec, conditioner=cond, num_frames_per_chunk=T, num_conditional_frames=tu["Tc"], num_steps=2)
    pipe.model._generate_initial_chunk.__func__      # chunk 0 path exists
    frame_u8 = ((frame.permute(1, 2, 0) + 1) * 127.5).clamp(0, 255).to(torch.uint8)
    video = pipe.image_to_video(frame_u8, num_frames=T)
    assert video.shape[0] == T and video.shape[3] == 3 and str(video.dtype) == "uint8"


def test_decoder_2d_vs_reference_golden(elem, golden_dir):
    """sgm Decoder (2-D; the arithmetic of the AutoencoderKL decoder used after the enhancer) vs the reference module's output."""
    from oracle import cases
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import Decoder2D, VaeConfig
    tv = cases.TINY_VAE
    dec = Decoder2D(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    dec.load_state_dict(init_by_name(dec.spec(), seed=7), device="cuda")
    out = dec(cases.tiny_vae_inputs()["z"][:2].cuda())
    report("2-D decoder vs reference", out, torch.load(os.path.join(golden_dir, "vae_dec2d_tiny.pt"))["out"])


def test_autoencoder_kl_2d_roundtrip_vs_oracle(elem):
    """AutoencoderKL2D (encoder + quant_conv mode, post_quant_conv + 2-D decoder, scaling 0.18215) vs the oracle pieces."""
    from oracle import cases, svd_oracle as O
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencoderKL2D, VaeConfig
    import torch.nn.functional as F
    tv = cases.TINY_VAE
    cfg, ocfg = VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]), O.VaeCfg(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])
    vae = AutoencoderKL2D(cfg)
    sd = init_by_name(vae.spec(), seed=12)
    vae.load_state_dict(sd, device="cuda")
    x = cases.tiny_vae_inputs()["x_enc"]
    lat = vae.encode_mode(x.cuda())
    with torch.no_grad():
        ref_lat = O.cond_frame_encode(sd, ocfg, x) * 0.18215
        z = F.conv2d(ref_lat / 0.18215, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        ref_img = O.vae_decoder_2d({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, ocfg, z)
    report("AutoencoderKL2D encode (mode)", lat, ref_lat)
    report("AutoencoderKL2D decode", vae.decode(ref_lat.cuda()), ref_img)


def test_clip_text_tower_vs_oracle(elem):
    """CLIPTextModel.last_hidden_state on the HIP kernels (causal mask as the score GEMM's residual, 77 -> 80 padded tokens) vs the
    oracle that is pinned against transformers' CLIPTextModel."""
    from oracle.clip_text_oracle import text_tower
    from streamingt2v_amd.clip_text import ClipTextConfig, CLIPTextTower
    from streamingt2v_amd.params import init_by_name
    cfg = ClipTextConfig(vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4)
    tower = CLIPTextTower(cfg)
    sd = init_by_name(tower.spec(), seed=13)
    tower.load_state_dict(sd, device="cuda")
    g = torch.Generator(); g.manual_seed(4)
    ids = torch.randint(0, 1000, (2, 77), generator=g)
    out = tower(ids)
    with torch.no_grad():
        ref = text_tower(sd, ids, cfg.heads)
    report("CLIP text tower", out.reshape(2, 77, 256), ref)
    with torch.no_grad():                              # clip_skip = 1: the enhancer's prompt path (pipeline_i2vgen_xl.py:246-260, default :645)
        ref1 = text_tower(sd, ids, cfg.heads, clip_skip=1)
    report("CLIP text tower, clip_skip 1", tower(ids, clip_skip=1).reshape(2, 77, 256), ref1)


def test_apm_spatial_video_transformer_vs_reference_golden(elem, golden_dir):
    """Appearance-preservation module (BASELINE north_star names it; config.yaml:115 ships it disabled): SpatialVideoTransformer with
    use_apm on a 17-token context on the HIP kernels -- Conv1d(17 -> 1) front as a GEMM, LayerNorm with the folded gate, and the temporal
    block's cross-attention to the 17 tokens through svd_attn_cross_d64 -- vs the UNMODIFIED reference module's output."""
    from oracle.cases import apm_inputs
    from streamingt2v_amd.params import Spec, init_by_name
    from streamingt2v_amd.video_model import SpatialVideoTransformer
    c = apm_inputs()
    C, T = c["C"], c["T"]
    svt = SpatialVideoTransformer("", C, 1024, use_apm=True)
    spec = Spec(); svt.spec(spec)
    svt.prepare(init_by_name(spec, seed=c["seed"]), "cuda")
    Fr, _, H, W = c["x"].shape
    tok = c["x"].permute(0, 2, 3, 1).reshape(Fr * H * W, C).to(ELEM).cuda().contiguous()
    ctx = c["context"].cuda()
    out = svt.forward(tok, ctx, ctx[::T].contiguous(), Fr, T, H, W).float().view(Fr, H, W, C).permute(0, 3, 1, 2)
    report("SpatialVideoTransformer + APM vs reference", out, torch.load(os.path.join(golden_dir, "apm_svt_tiny.pt"))["out"])


def test_groupnorm_sums_and_sequence_parallel_path_on_one_rank(tiny):
    """The sequence-parallel code path on the REAL kernels: svd_groupnorm_sums / svd_groupnorm_stats_from_sums vs torch, and the SP
    forward over a one-rank RCCL group (layout changes degenerate, every SP branch of the networks runs: pooled GroupNorms from summed
    statistics, pixel-layout temporal operators, gathered CAM keys / values and network output) == the plain forward."""
    import socket
    import torch.distributed as dist
    from streamingt2v_amd import ops, parallel
    g = torch.Generator(); g.manual_seed(3)
    Fr, pix, C, fps = 6, 40, 320, 3
    x = (torch.randn(Fr * pix, C, generator=g) * 1.5 + 0.5).to(ELEM).cuda()
    sums = ops.groupnorm_sums(x, Fr, pix, fps)
    v = x.double().view(Fr // fps, fps * pix, 32, C // 32)
    ref = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)
    assert torch.allclose(sums, ref, rtol=1e-5, atol=1e-3), (sums - ref).abs().max()
    gam, bet = torch.randn(C, generator=g).cuda() * 0.1 + 1, torch.randn(C, generator=g).cuda() * 0.1
    a = ops.groupnorm_apply_sums(x, Fr, pix, gam, bet, 1e-5, sums, float(fps * pix * (C // 32)), frames_per_stat=fps, silu=True)
    b = ops.groupnorm(x, Fr, pix, gam, bet, 1e-5, frames_per_stat=fps, silu=True)
    assert (a.float() - b.float()).abs().max().item() <= 2e-2 * (1.0 if ELEM == torch.bfloat16 else 0.125)
    if not dist.is_initialized():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    tu, wrap = tiny["tu"], tiny["wrap"]
    inp = _cuda(tiny["cases"].tiny_wrapper_inputs())
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    kw = dict(batch_size=2, num_video_frames=tu["T"], image_only_indicator=torch.zeros(2, tu["T"], device="cuda"), ctrl_frames=inp["ctrl_frames"])
    ref = wrap.forward(inp["x"], inp["t"], c, **kw)
    try:
        wrap.sp = parallel.SeqParallel(None)
        got = wrap.forward(inp["x"], inp["t"], c, **kw)
    finally:
        wrap.sp = None
    report("sequence-parallel path (1 rank) vs plain forward", got, ref, rel_tol=1e-2)
