"""CPU suite: the HOST LOGIC of the denoiser networks (streamingt2v_amd/video_model.py, wrappers.py, sampling.py) against the CPU oracle,
with the HIP launchers replaced by fp32 torch statements of the same operators (tests/svd_shim.py).  What this pins without a GPU: weight
packing (conv / temporal / GEGLU interleave / fused q|k|v), epilogue bookkeeping (bias, per-frame vectors, residual, alpha blend), the
ControlNet slicing of StreamingWrapper (first Tc frames per CFG half, CLIP token 0), the CAM wiring (un-merged middle block), the K3
shortcut (1-token cross-attention == to_out(to_v(ctx))), and the fused sampler step.  fp32 on both sides: agreement to rounding."""
import pytest
import torch

from tests import svd_shim


@pytest.fixture()
def tiny(monkeypatch):
    svd_shim.install(monkeypatch)
    from oracle import cases, svd_oracle as O
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.manual_seed(0)
    tu = cases.TINY_UNET
    cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                     conditioning_embedding_out_channels=tu["cond_embed"])
    unet, cn = VideoUNet(cfg), ControlNet(cfg)
    sd_u, sd_c = init_by_name(unet.spec(), seed=1), init_by_name(cn.spec(), seed=2)
    unet.load_state_dict(sd_u, device="cpu")
    cn.load_state_dict(sd_c, device="cpu")
    ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                 cond_embed_channels=tu["cond_embed"])
    return dict(wrap=StreamingWrapper(unet, cn, tu["Tc"]), unet=unet, sd_u=sd_u, sd_c=sd_c, ocfg=ocfg, tu=tu, cases=cases, O=O)


def test_streaming_wrapper_host_logic_vs_oracle_and_reference_golden(tiny, golden_dir):
    import os
    tu, O = tiny["tu"], tiny["O"]
    inp = tiny["cases"].tiny_wrapper_inputs()
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    with torch.no_grad():
        out = tiny["wrap"].forward(inp["x"], inp["t"], c, batch_size=2, num_video_frames=tu["T"], image_only_indicator=torch.zeros(2, tu["T"]),
                                   ctrl_frames=inp["ctrl_frames"])
        ref = O.streaming_wrapper(tiny["sd_u"], tiny["sd_c"], tiny["ocfg"], inp["x"], inp["t"], c, 2, tu["T"], tu["Tc"], inp["ctrl_frames"])
    gold = torch.load(os.path.join(golden_dir, "wrapper_tiny.pt"))["out"]          # the unmodified reference's output
    assert (out - ref).abs().max().item() < 2e-4, (out - ref).abs().max()
    assert (out - gold).abs().max().item() < 2e-4, (out - gold).abs().max()


def test_control_frames_of_the_wrong_size_raise(tiny):
    """A decoder that does not upsample 8x hands over control frames the ControlNet cannot use: a clear error, not an out-of-bounds read."""
    tu = tiny["tu"]
    inp = tiny["cases"].tiny_wrapper_inputs()
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    bad = inp["ctrl_frames"][..., ::4, ::4].contiguous()
    with pytest.raises(ValueError, match="control frames"):
        tiny["wrap"].forward(inp["x"], inp["t"], c, batch_size=2, num_video_frames=tu["T"], image_only_indicator=torch.zeros(2, tu["T"]), ctrl_frames=bad)


def test_fused_sampler_vs_oracle(tiny):
    from streamingt2v_amd.sampling import EulerEDMSampler
    tu, O = tiny["tu"], tiny["O"]
    sin = tiny["cases"].tiny_sampler_inputs()
    inp = tiny["cases"].tiny_wrapper_inputs()
    T = tu["T"]
    with torch.no_grad():
        z = EulerEDMSampler(num_steps=2, num_frames=T)(tiny["wrap"], sin["noise"].clone(), sin["c"], sin["uc"], batch_size=2, num_video_frames=T,
                                                       ctrl_frames=inp["ctrl_frames"])
        net = lambda a, cn_, cc: O.streaming_wrapper(tiny["sd_u"], tiny["sd_c"], tiny["ocfg"], a, cn_, cc, 2, T, tu["Tc"], inp["ctrl_frames"])
        ref = O.euler_edm_sample(net, sin["noise"].clone(), sin["c"], sin["uc"], 2, T)
    assert ((z - ref).abs().max() / ref.abs().max()).item() < 1e-4


def test_two_videos_through_one_wrapper_do_not_share_control_state(tiny):
    """the control-frame caches must key on the tensor OBJECT.  Two videos whose control tensors have the same shape, version
    and (after the first is freed) possibly the same address must each see their own embedding."""
    tu = tiny["tu"]
    inp = tiny["cases"].tiny_wrapper_inputs()
    c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
    kw = dict(batch_size=2, num_video_frames=tu["T"], image_only_indicator=torch.zeros(2, tu["T"]))
    g = torch.Generator(); g.manual_seed(5)
    outs = []
    with torch.no_grad():
        for _ in range(2):
            ctrl = torch.rand(inp["ctrl_frames"].shape, generator=g) * 2 - 1
            outs.append((ctrl.clone(), tiny["wrap"].forward(inp["x"], inp["t"], c, ctrl_frames=ctrl, **kw)))
            del ctrl                                                   # freed: the next allocation may reuse the address
        from streamingt2v_amd.video_model import ControlNet
        from streamingt2v_amd.wrappers import StreamingWrapper
        for ctrl, got in outs:
            cn = ControlNet(tiny["unet"].cfg)
            cn.load_state_dict(tiny["sd_c"], device="cpu")
            fresh = StreamingWrapper(tiny["unet"], cn, tu["Tc"]).forward(inp["x"], inp["t"], c, ctrl_frames=ctrl, **kw)
            assert torch.equal(got, fresh)


def test_per_chunk_context_constants_follow_the_context_tensor(tiny):
    """The one-token cross-attention constants (video_model._attn2_const) and the derived context tensors are cached per context OBJECT: the
    same object twice must hit (bit-equal output), a new object with other values -- or the same object modified in place -- must not."""
    tu = tiny["tu"]
    inp = tiny["cases"].tiny_wrapper_inputs()
    kw = dict(batch_size=2, num_video_frames=tu["T"], image_only_indicator=torch.zeros(2, tu["T"]), ctrl_frames=inp["ctrl_frames"])
    from streamingt2v_amd.video_model import ControlNet, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper

    def fresh(c):
        u, cn = VideoUNet(tiny["unet"].cfg), ControlNet(tiny["unet"].cfg)
        u.load_state_dict(tiny["sd_u"], device="cpu"); cn.load_state_dict(tiny["sd_c"], device="cpu")
        return StreamingWrapper(u, cn, tu["Tc"]).forward(inp["x"], inp["t"], c, **kw)

    with torch.no_grad():
        c1 = {k: inp[k].float().contiguous() for k in ("concat", "crossattn", "vector")}
        a = tiny["wrap"].forward(inp["x"], inp["t"], c1, **kw)
        b = tiny["wrap"].forward(inp["x"], inp["t"], c1, **kw)                       # same objects: cache hits
        assert torch.equal(a, b) and torch.equal(a, fresh(c1))
        c2 = dict(c1, crossattn=(c1["crossattn"] * -0.5 + 0.25).contiguous())         # the next video's context
        d = tiny["wrap"].forward(inp["x"], inp["t"], c2, **kw)
        assert torch.equal(d, fresh(c2)) and not torch.equal(d, a)
        c2["crossattn"].mul_(2.0)                                                      # in place: same object, new version
        e = tiny["wrap"].forward(inp["x"], inp["t"], c2, **kw)
        assert torch.equal(e, fresh(c2)) and not torch.equal(e, d)


def test_apm_block_host_logic_vs_reference_golden(monkeypatch, golden_dir):
    """Appearance-preservation module (use_apm: true; attention.py:596-620): SpatialVideoTransformer on a 17-token context -- the spatial
    block's Conv1d + LayerNorm + gate front (as a GEMM on an im2col of the CLIP axis) and the temporal block's real cross-attention to the
    17 tokens -- against the output of the UNMODIFIED reference module (oracle/make_golden_apm.py)."""
    import os
    svd_shim.install(monkeypatch)
    from oracle.cases import apm_inputs
    from streamingt2v_amd.params import Spec, init_by_name
    from streamingt2v_amd.video_model import SpatialVideoTransformer
    c = apm_inputs()
    C, T = c["C"], c["T"]
    svt = SpatialVideoTransformer("", C, 1024, use_apm=True)
    spec = Spec(); svt.spec(spec)
    svt.prepare(init_by_name(spec, seed=c["seed"]), "cpu")
    Fr, _, H, W = c["x"].shape
    tok = c["x"].permute(0, 2, 3, 1).reshape(Fr * H * W, C).contiguous()
    with torch.no_grad():
        out = svt.forward(tok, c["context"], c["context"][::T].contiguous(), Fr, T, H, W)
    out = out.view(Fr, H, W, C).permute(0, 3, 1, 2)
    gold = torch.load(os.path.join(golden_dir, "apm_svt_tiny.pt"))["out"]
    assert (out - gold).abs().max().item() < 5e-4, (out - gold).abs().max()
    # a one-token context through a use_apm block takes the plain path (attention.py:614: context.shape[1] > 1)
    plain = SpatialVideoTransformer("", C, 1024, use_apm=False)
    spec2 = Spec(); plain.spec(spec2)
    sd = init_by_name(spec, seed=c["seed"])
    plain.prepare(sd, "cpu")
    one = c["context"][:, :1].contiguous()
    from streamingt2v_amd import ops
    ctx1, tctx1 = ops.to_bf16(one[:, 0].contiguous()), ops.to_bf16(one[::T, 0].contiguous())
    with torch.no_grad():
        assert torch.equal(svt.forward(tok, ctx1, tctx1, Fr, T, H, W), plain.forward(tok, ctx1, tctx1, Fr, T, H, W))


def test_initial_chunk_follows_the_diffusers_pipeline(monkeypatch):
    """Row N2: chunk 0 as the reference computes it -- diffusers' StableVideoDiffusionPipeline.__call__(image, decode_chunk_size=8)
    (streaming_svd.py:388-394) -- restated IN DIFFUSERS' OWN FORMULATION in oracle/svd_pipeline_oracle.py (EulerDiscreteScheduler with Karras sigmas
    from a float64 numpy ramp, scale_model_input, v-prediction pred_original_sample, init_noise_sigma, guidance linspace(1, 3), 0.02 * randn
    augmentation, `_resize_with_antialiasing`, un-scaled image latents, uint8 PIL frames) against the product's path in ITS formulation:
    SVDConditioner.first_chunk -> StreamingSVD._generate_initial_chunk (EDM sampler with VScaling, fused step) -> quantize_like_pil, with the
    stock-weights slot (`set_initial_model`) in use.  fp32 on both sides (shim launchers): latents to rounding, frames to one uint8 level."""
    svd_shim.install(monkeypatch)
    from oracle import cases, svd_oracle as O, svd_pipeline_oracle as PO
    from streamingt2v_amd.conditioner import SVDConditioner
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    tu, tv = cases.TINY_UNET, cases.TINY_VAE
    T, h, w, steps = 5, 8, 8, 3
    ucfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"], controlnet_mode=False)
    unet = VideoUNet(ucfg)
    sd_u = init_by_name(unet.spec(), seed=21)
    unet.load_state_dict(sd_u, device="cpu")
    dec = VideoDecoder(VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]))
    sd_d = init_by_name(dec.spec(), seed=22)
    dec.load_state_dict(sd_d, device="cpu")
    ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"])
    vcfg = O.VaeCfg(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])
    up = 2 ** (len(tv["ch_mult"]) - 1)
    g = torch.Generator().manual_seed(77)
    image01 = torch.rand(1, 3, 8 * h, 8 * w, generator=g)                       # "the PIL image"
    aug = torch.randn(1, 3, 8 * h, 8 * w, generator=g)
    lat = torch.randn(1, T, 4, h, w, generator=g)
    # ---- the product: a StreamingSVD whose own networks must NOT be touched for chunk 0 once the stock slot is filled
    class Boom:
        def __getattr__(self, k):
            raise AssertionError("chunk 0 must run on the stock-weights slot")
    svd = StreamingSVD(Boom(), Boom(), EulerEDMSampler(num_steps=30, num_frames=T), num_conditional_frames=tu["Tc"])
    svd.initial_num_steps = steps
    cond = SVDConditioner(cases.fake_clip_embed, cases.fake_cond_encode, num_frames=T)
    svd.set_initial_model(StreamingWrapper(unet, None, tu["Tc"]), AutoencodingEngineDecoder(dec), cond)
    with torch.no_grad():
        c, uc = svd.initial_conditioning(None, image01[0] * 2 - 1)              # draws 0.02 * randn itself ...
        c2, uc2 = cond.first_chunk(image01[0] * 2 - 1, aug_noise=aug)           # ... here with the oracle's draw
        assert c["concat"].shape == c2["concat"].shape and not torch.equal(c["concat"], c2["concat"]) and torch.equal(c["crossattn"], c2["crossattn"])
        zs, dec_fs = [], svd.decode_first_stage
        svd.decode_first_stage = lambda z, **kw: (zs.append(z.clone()), dec_fs(z, **kw))[1]          # the latents the product hands its decoder
        frames = svd.quantize_like_pil(svd._generate_initial_chunk(c2, uc2, lat[0]))
        z_prod = zs[0]
        # ---- diffusers' formulation on the oracle networks
        unet_d = PO.sgm_unet_as_diffusers(lambda x, t, ctx, y: O.video_unet(sd_u, ocfg, x, t, ctx, y, T), T)
        u8, z_ref = PO.svd_pipeline_call(image01, cases.fake_clip_embed, cases.fake_cond_encode, unet_d,
                                         lambda z, num_frames: O.video_decoder(sd_d, vcfg, z, num_frames), aug_noise=aug, latents=lat,
                                         num_frames=T, num_inference_steps=steps, decode_chunk_size=8)
    ref = PO.frames_back_to_float(u8)
    assert frames.shape == ref.shape == (T, 3, up * h, up * w)
    ez = (z_prod - z_ref[0]).abs().max().item() / z_ref.abs().max().item()
    lv = ((frames - ref).abs() * 127.5).round()
    print(f"[chunk 0 vs the diffusers-formulation oracle] latents max rel err {ez:.2e}; uint8 frames: {100 * (lv > 0).float().mean():.3f} % of bytes differ, max {int(lv.max())} level")
    assert ez < 2e-4, ez
    assert lv.max().item() <= 1 and (lv > 0).float().mean().item() < 5e-3
    # the CLIP resize of chunk 0 is diffusers' always-blurring copy of kornia's: identical on a down-scale, and when nothing shrinks its sigma of
    # 0.001 makes the 3-tap blur an identity
    x = torch.rand(1, 3, 300, 500, generator=g) * 2 - 1
    assert torch.allclose(SVDConditioner.clip_preprocess(x, always_blur=True), SVDConditioner.clip_preprocess(x), atol=1e-6)
    from streamingt2v_amd.conditioner import kornia_resize_antialias
    small = torch.rand(1, 3, 100, 100, generator=g)
    assert torch.allclose(kornia_resize_antialias(small, (224, 224), always_blur=True), PO.resize_with_antialiasing(small, (224, 224)), atol=1e-6)
    assert torch.allclose(kornia_resize_antialias(small, (224, 224)), PO.resize_with_antialiasing(small, (224, 224)), atol=1e-6)


def test_svd_pipeline_mirror_takes_the_place_of_the_reference_pipeline(monkeypatch):
    """Row N2, reference side: `dropin.install_svd_pipeline(model)` replaces the reference's `self.svd_pipeline` (diffusers
    StableVideoDiffusionPipeline) by a mirror built from the PIPELINE'S OWN unet / vae / image_encoder state dicts (diffusers names -> sgm
    names, strict), and `self.svd_pipeline(image, decode_chunk_size=8).frames[0]` (streaming_svd.py:390) then yields the PIL frames the
    restated pipeline call (oracle/svd_pipeline_oracle.py, diffusers' formulation) yields on the same two random draws."""
    import types
    import numpy as np
    import PIL.Image
    svd_shim.install(monkeypatch)
    pytest.importorskip("transformers")
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle import cases, svd_oracle as O, svd_pipeline_oracle as PO
    from streamingt2v_amd import dropin
    from streamingt2v_amd.diffusers_keys import sgm_temporal_decoder_key_to_diffusers as dkey, sgm_unet_key_to_diffusers as ukey
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    tu = cases.TINY_UNET
    T, H, W, steps = 5, 64, 64, 3
    ucfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"], controlnet_mode=False)
    sd_u = init_by_name(VideoUNet(ucfg).spec(), seed=21)
    vcfg = VaeConfig(32, (1, 1, 1, 2), 1)                                  # 4 levels: latents at 1 / 8 of the image like the shipped VAE
    sd_d, sd_e = init_by_name(VideoDecoder(vcfg).spec(), seed=22), init_by_name(CondFrameEncoder(vcfg).spec(), seed=23)
    vae_dif = {}
    for k, v in sd_d.items():
        nk = dkey(k, 4)
        vae_dif[nk] = v[:, :, 0, 0] if "attentions.0.to_" in nk and v.dim() == 4 else v
    for k, v in sd_e.items():                                              # the encoder half in diffusers' names
        part, _, r = k.partition(".")
        if part == "quant_conv":
            vae_dif[k] = v; continue
        r = r.replace("norm_out.", "conv_norm_out.").replace("nin_shortcut.", "conv_shortcut.").replace("mid.block_1.", "mid_block.resnets.0.").replace("mid.block_2.", "mid_block.resnets.1.")
        if r.startswith("mid.attn_1."):
            r = r.replace("mid.attn_1.", "mid_block.attentions.0.").replace("proj_out.", "to_out.0.").replace("norm.", "group_norm.")
            for n in "qkv":
                r = r.replace(f"attentions.0.{n}.", f"attentions.0.to_{n}.")
            v = v[:, :, 0, 0] if v.dim() == 4 else v
        if r.startswith("down."):
            _, i, kind, rest = r.split(".", 3)
            r = f"down_blocks.{i}.resnets.{rest}" if kind == "block" else f"down_blocks.{i}.downsamplers.0.{rest}"
        vae_dif["encoder." + r] = v

    class Mod:                                                             # what the mirror reads from a diffusers module: state_dict() and config
        def __init__(self, sd, config):
            self._sd, self.config = sd, config
        def state_dict(self):
            return self._sd
    boc = [320 * m for m in tu["channel_mult"]]
    down = ["CrossAttnDownBlockSpatioTemporal" if (2 ** i) in tu["attention_resolutions"] else "DownBlockSpatioTemporal" for i in range(len(boc))]
    icfg = dict(hidden_size=320, intermediate_size=1280, num_hidden_layers=1, num_attention_heads=4, image_size=224, patch_size=14, projection_dim=1024)
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_act="gelu", **icfg))
    hf.config_dict = icfg
    pipe = types.SimpleNamespace(unet=Mod({ukey(k, ucfg.num_res_blocks): v for k, v in sd_u.items()},
                                          dict(block_out_channels=boc, layers_per_block=tu["num_res_blocks"], down_block_types=down, num_frames=T)),
                                 vae=Mod(vae_dif, dict(block_out_channels=[32, 32, 32, 64], layers_per_block=1)),
                                 image_encoder=Mod({k: v for k, v in hf.state_dict().items()}, icfg))
    model = types.SimpleNamespace(svd_pipeline=pipe, inference_params=types.SimpleNamespace(num_conditional_frames=tu["Tc"]))
    mirror = dropin.install_svd_pipeline(model, device="cpu")
    assert model.svd_pipeline is mirror and mirror.num_frames == T and mirror.conditioner.clip.cfg.layers == 1
    # the conditioner's two towers are replaced by the linear stand-ins ON BOTH SIDES (the towers themselves are pinned elsewhere; on CPU only their
    # construction from the pipeline's weights is exercised): what is compared is the call
    mirror.conditioner.clip, mirror.conditioner.enc = cases.fake_clip_embed, cases.fake_cond_encode
    rs = np.random.RandomState(3)
    pil = PIL.Image.fromarray((rs.rand(H, W, 3) * 255).astype("uint8"))
    with torch.no_grad():
        out = model.svd_pipeline(pil, decode_chunk_size=8, height=H, width=W, num_inference_steps=steps, generator=torch.Generator().manual_seed(5))
        frames = out.frames[0]
        assert len(frames) == T and isinstance(frames[0], PIL.Image.Image) and frames[0].size == (W, H)
        g = torch.Generator().manual_seed(5)                               # the pipeline's two draws, in its order
        aug, lat = torch.randn(1, 3, H, W, generator=g), torch.randn(T, 4, H // 8, W // 8, generator=g)
        image01 = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1)[None].float() / 255.0
        ocfg = O.Cfg(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"])
        ovae = O.VaeCfg(32, (1, 1, 1, 2), 1)
        unet_d = PO.sgm_unet_as_diffusers(lambda x, t, ctx, y: O.video_unet(sd_u, ocfg, x, t, ctx, y, T), T)
        u8, _ = PO.svd_pipeline_call(image01, cases.fake_clip_embed, cases.fake_cond_encode, unet_d, lambda z, num_frames: O.video_decoder(sd_d, ovae, z, num_frames),
                                     aug_noise=aug, latents=lat[None], num_frames=T, num_inference_steps=steps, decode_chunk_size=8)
    got = torch.from_numpy(np.stack([np.asarray(f) for f in frames]))
    lv = (got.int() - u8.int()).abs()
    print(f"[svd_pipeline mirror vs the restated pipeline call] uint8 frames: {100 * (lv > 0).float().mean():.3f} % of bytes differ, max {int(lv.max())} level")
    assert got.shape == u8.shape and lv.max().item() <= 1 and (lv > 0).float().mean().item() < 5e-3
