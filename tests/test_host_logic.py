"""CPU suite, part 3: host-side logic that needs no GPU -- state-dict specs, weight packing, schedules, AR bookkeeping."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import svd_oracle as O


def test_full_size_specs():
    from streamingt2v_amd.temporal_ae import VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, VideoUNet
    u, c, d = VideoUNet().spec(), ControlNet().spec(), VideoDecoder().spec()
    # 1 593.5 M / 673.0 M / 63.6 M parameters, 1571 + 657 tensors
    assert len(u) == 1571 and abs(u.numel() / 1e6 - 1593.5) < 0.05
    assert len(c) == 657 and abs(c.numel() / 1e6 - 673.0) < 0.05
    assert abs(d.numel() / 1e6 - 63.6) < 0.05
    assert len(set(u.names())) == len(u)
    assert "cross_attention_merger_mid_block.temporal_transformer.attention.to_out.0.bias" in u.names()
    assert "controlnet_cond_embedding.norms.5.weight" in c.names()
    assert "conv_out.time_mix_conv.weight" in d.names()


def test_strict_state_dict_check():
    import pytest
    from streamingt2v_amd.params import Spec, check_state_dict, init_by_name
    s = Spec(); s.add("a.weight", 4, 3); s.add("a.bias", 4)
    sd = init_by_name(s, seed=5)
    check_state_dict(s, sd)
    assert torch.equal(sd["a.weight"], init_by_name(s, seed=5)["a.weight"])          # deterministic by name
    assert not torch.equal(sd["a.weight"], init_by_name(s, seed=6)["a.weight"])
    with pytest.raises(RuntimeError):
        check_state_dict(s, {"a.weight": sd["a.weight"]})
    with pytest.raises(RuntimeError):
        check_state_dict(s, dict(sd, extra=torch.zeros(1)))
    with pytest.raises(RuntimeError):
        check_state_dict(s, {"a.weight": torch.zeros(3, 4), "a.bias": sd["a.bias"]})


def test_weight_packing_semantics():
    from streamingt2v_amd.video_model import pack_conv3x3, pack_geglu, pack_tconv3
    g = torch.Generator(); g.manual_seed(0)
    # conv: packed weight times im2col(ky,kx,c) == F.conv2d
    x = torch.randn(2, 5, 6, 7, generator=g); w = torch.randn(4, 5, 3, 3, generator=g)
    cols = F.unfold(x, 3, padding=1).view(2, 5, 9, 42).permute(0, 3, 2, 1).reshape(2 * 42, 45)     # (ky kx) c
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1).reshape(2 * 42, 4)
    assert torch.allclose(cols @ pack_conv3x3(w).t(), ref, atol=1e-4)
    wp = pack_conv3x3(w, cin_pad=32, cout_pad=8)
    assert wp.shape == (8, 9 * 32) and torch.equal(wp[4:], torch.zeros(4, 288))
    # temporal conv
    xt = torch.randn(1, 5, 4, 3, 1, generator=g); wt = torch.randn(6, 5, 3, 1, 1, generator=g)
    ref = F.conv3d(xt, wt, padding=(1, 0, 0))[0, :, :, :, 0]                                         # c t p
    xp = F.pad(xt[0, :, :, :, 0], (0, 0, 1, 1))                                                      # c t+2 p
    cols = torch.stack([xp[:, k:k + 4] for k in range(3)], 0).permute(2, 3, 0, 1).reshape(12, 15)    # (t p) (kt c)
    assert torch.allclose(cols @ pack_tconv3(wt).t(), ref.permute(1, 2, 0).reshape(12, 6), atol=1e-4)
    # GEGLU interleave: value block b at rows [64b, 64b+32), its gate at [64b+32, 64b+64)
    w = torch.randn(128, 8, generator=g); b = torch.randn(128, generator=g)
    wp, bp = pack_geglu(w, b)
    assert torch.equal(wp[0:32], w[0:32]) and torch.equal(wp[32:64], w[64:96])
    assert torch.equal(wp[64:96], w[32:64]) and torch.equal(wp[96:128], w[96:128]) and torch.equal(bp[32:64], b[64:96])


def test_schedule_and_scaling_match_oracle():
    from streamingt2v_amd.sampling import AlignYourSteps, EulerEDMSampler, VScalingWithEDMcNoise
    for n in (2, 4, 25, 30):
        assert np.array_equal(AlignYourSteps()(n), O.ays_sigmas(n).numpy())
    import os
    from streamingt2v_amd.sampling import EDMDiscretization
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_tiny.pt"))
    np.testing.assert_allclose(EDMDiscretization()(25), gold["edm25"].double().numpy(), rtol=2e-6)     # reference's EDMDiscretization
    np.testing.assert_allclose(EDMDiscretization()(25), O.edm_sigmas(25).double().numpy(), rtol=2e-6)
    assert abs(EDMDiscretization()(25)[0] - 700.0) < 1e-3 and abs(EDMDiscretization()(25)[24] - 0.002) < 1e-6   # fp32 pow, like the reference
    s = EulerEDMSampler(num_steps=30, num_frames=25)
    assert torch.equal(s.guider.scale, torch.linspace(1.5, 3.0, 25))
    for sg in (700.0, 1.0, 0.002):
        ours = VScalingWithEDMcNoise()(sg)
        ref = [float(v) for v in O.vscaling_edm(torch.tensor(sg, dtype=torch.float64))]
        np.testing.assert_allclose(ours, ref, rtol=1e-12)


def test_autoregressive_bookkeeping():
    """AR outer loop: ctrl frames = last 7 of the previous chunk, anchor = chunk0[6], 18 new frames kept per chunk."""
    from streamingt2v_amd.streaming_svd import StreamingSVD

    class FakeSampler:
        class guider:
            num_frames = 25

    seen = []

    class Model(StreamingSVD):
        def _generate_conditional_output(self, c, uc, ctrl_frames, noise, num_steps=None):
            seen.append((c["anchor"], ctrl_frames.clone()))
            k = len(seen)
            return torch.full((25, 3, 2, 2), float(k)) + torch.arange(25)[:, None, None, None] / 100.0

    m = Model(None, None, FakeSampler(), num_conditional_frames=7)
    init = torch.arange(25)[:, None, None, None].float().expand(25, 3, 2, 2) / 100.0
    out = m._autoregressive_generation(init, lambda a: ({"anchor": a}, {}), 3, [None] * 3)
    assert out.shape[0] == 25 + 3 * 18                       # ceil((N-25)/18) chunks, inference_i2v.py:179-184
    assert all(torch.equal(a, init[6]) for a, _ in seen)      # anchor fixed to the 7th frame of chunk 0
    assert torch.allclose(seen[0][1][0], init[-7:], atol=2e-7)   # first ctrl frames = tail of chunk 0 (through the reference's range round trip)
    assert torch.allclose(seen[1][1][0, :, 0, 0, 0], 1.0 + torch.arange(18, 25) / 100.0)   # then tail of chunk 1
    assert torch.allclose(out[25:43, 0, 0, 0], 1.0 + torch.arange(7, 25) / 100.0)          # overlap frames dropped


def test_image_to_video_chunk_arithmetic():
    """inference_i2v.py:179-190: 100 diffusion-stage frames = chunk 0 (25) + ceil((100-25)/18) = 5 AR chunks, cut to 100."""
    from streamingt2v_amd.streaming_svd import StreamingSVD

    class FakeSampler:
        cfg_exchange = None
        class guider:
            num_frames = 25

    calls = []

    class Model(StreamingSVD):
        def _generate_initial_chunk(self, c, uc, noise, **kw):
            calls.append("init")
            return torch.zeros(25, 3, 2, 2)
        def _generate_conditional_output(self, c, uc, ctrl_frames, noise, num_steps=None):
            calls.append("ar")
            return torch.ones(25, 3, 2, 2) * len(calls)

    m = Model(None, None, FakeSampler(), num_conditional_frames=7)
    out = m.image_to_video(lambda a: ({}, {}), torch.zeros(3, 2, 2), 100, [None] * 6)
    assert calls == ["init"] + ["ar"] * 5 and out.shape[0] == 100
    out = m.image_to_video(lambda a: ({}, {}), torch.zeros(3, 2, 2), 25, [None])
    assert out.shape[0] == 25


def test_ddim_schedule_host_tables_match_oracle():
    """enhance.DDIMSchedule (host tables feeding svd_ddim_cfg_step) vs the oracle's DDIM restatement: timesteps, SDEdit start, alphas."""
    import torch
    from oracle.i2vgen_oracle import DDIM
    from streamingt2v_amd.enhance import DDIMSchedule
    ours, ora = DDIMSchedule(), DDIM()
    for n in (30, 10, 50):
        ora.set_timesteps(n)
        assert ours.set_timesteps(n) == ora.timesteps.tolist()
    ora.set_timesteps(30)
    assert ours.get_timesteps(30, 0.97) == ora.timesteps.tolist()[1:] and len(ours.get_timesteps(30, 0.97)) == 29
    for t in ours.get_timesteps(30, 0.97):
        a_t, a_prev = ours.alphas(t)
        prev = t - 1000 // 30
        assert abs(a_t - ora.alphas_cumprod[t].item()) < 1e-7
        assert abs(a_prev - (ora.alphas_cumprod[prev].item() if prev >= 0 else ora.final_alpha_cumprod.item())) < 1e-7
    g = torch.Generator(); g.manual_seed(0)
    x, n = torch.randn(1, 4, 3, 5, 5, generator=g), torch.randn(1, 4, 3, 5, 5, generator=g)
    assert torch.allclose(ours.add_noise(x, n, 958), ora.add_noise(x, n, 958), atol=1e-6)


def test_i2v_spec_parameter_count():
    """The enhancer mirror declares the reference architecture: 1511 tensors / 1420.5 M parameters at full size."""
    import math
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    spec = I2VGenXLUNet(I2VConfig()).spec()
    assert len(spec) == 1511
    assert abs(sum(math.prod(s) for _, s in spec) / 1e6 - 1420.5) < 0.1


def test_pipeline_front_end_chunk_arithmetic_and_windows():
    """streamingt2v_amd.pipeline vs the reference's expressions (inference_i2v.py:182-186, i2v_enhance_interface.py:88-113)."""
    import math
    from streamingt2v_amd import pipeline as P
    for n in (25, 26, 43, 100, 101, 200):
        assert P.num_autoregressive_generations(n) == max(0, math.ceil((n - 25) / (25 - 7)))
    for n, chunk, overlap in ((100, 38, 12), (100, 50, 0), (38, 38, 12), (37, 38, 12), (115, 38, 12), (200, 38, 12)):
        video = list(range(n))
        ref_chunks = [video[i:i + chunk] for i in range(0, len(video), chunk - overlap) if len(video[i:i + chunk]) == chunk]
        starts, max_idx = P.enhance_windows(n, chunk, overlap)
        assert starts == [c[0] for c in ref_chunks]
        assert max_idx == ((chunk - overlap) * (len(ref_chunks) - 1) + chunk if ref_chunks else 0)
    assert P.DEFAULTS["chunk_size"] == 38 and P.DEFAULTS["overlap_size"] == 12 and P.DEFAULTS["num_frames"] == 200


def test_pipeline_checkpoint_key_map(tmp_path):
    """A safetensors checkpoint with the reference's key prefixes loads strictly into the three mirrors; a missing key fails."""
    import pytest
    import torch
    from safetensors.torch import save_file
    from oracle import cases
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    tu, tv = cases.TINY_UNET, cases.TINY_VAE
    ucfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                      channel_mult=tu["channel_mult"], conditioning_embedding_out_channels=tu["cond_embed"])
    vcfg = VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])
    sd = {}
    for prefix, mod, seed in ((P.CKPT_PREFIXES["unet"], VideoUNet(ucfg), 1), (P.CKPT_PREFIXES["controlnet"], ControlNet(ucfg), 2),
                              (P.CKPT_PREFIXES["decoder"], VideoDecoder(vcfg), 3)):
        sd.update({prefix + k: v.contiguous() for k, v in init_by_name(mod.spec(), seed=seed).items()})
    sd["conditioner.embedders.0.dummy"] = torch.zeros(1)            # keys of parts that stay on the reference side are ignored
    path = str(tmp_path / "model.safetensors")
    save_file(sd, path)
    unet, cnet, dec = P.load_streamingsvd_checkpoint(path, device="cpu", unet_cfg=ucfg, vae_cfg=vcfg)
    assert unet.prepared and cnet.prepared
    sd.pop(P.CKPT_PREFIXES["unet"] + "out.2.weight")
    with pytest.raises(RuntimeError):
        P.load_streamingsvd_checkpoint(sd, device="cpu", unet_cfg=ucfg, vae_cfg=vcfg)
    pipe = P.StreamingPipeline(unet, cnet, dec)
    with pytest.raises(NotImplementedError):
        pipe.image_to_video(torch.zeros(64, 64, 3, dtype=torch.uint8), 25)
    with pytest.raises(NotImplementedError):
        pipe.interpolate_video([], 10)


def test_clip_vision_spec_is_vit_h_14():
    """Full-size tower spec: open_clip ViT-H-14 visual (632 M parameters, 257 x 1280 positions, 1024-d embedding)."""
    import math
    from streamingt2v_amd.clip_vision import OpenCLIPVisionTower
    spec = dict(OpenCLIPVisionTower().spec())
    assert spec["visual.positional_embedding"] == (257, 1280) and spec["visual.proj"] == (1280, 1024)
    assert spec["visual.transformer.resblocks.31.attn.in_proj_weight"] == (3840, 1280)
    assert abs(sum(math.prod(s) for s in spec.values()) / 1e6 - 632.1) < 0.5


def test_svd_conditioner_assembly_with_stub_towers():
    """SVDConditioner: shapes, uc zeroing, vector = [fps_id | motion_bucket_id | cond_aug] sinusoids, uniform cond-aug noise."""
    import torch
    from oracle import svd_oracle as O
    from streamingt2v_amd.conditioner import SVDConditioner, sinusoid
    seen = {}

    def clip(x):
        seen["clip"] = x
        return torch.ones(x.shape[0], 1024)

    def enc(x):
        seen["enc"] = x
        return torch.full((x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8), 0.5)

    frame = torch.rand(3, 64, 96) * 2 - 1
    c, uc = SVDConditioner(clip, enc, num_frames=5, generator=torch.Generator().manual_seed(1))(frame)
    assert c["crossattn"].shape == (5, 1, 1024) and c["concat"].shape == (5, 4, 8, 12) and c["vector"].shape == (5, 768)
    assert uc["crossattn"].abs().sum() == 0 and uc["concat"].abs().sum() == 0 and torch.equal(uc["vector"], c["vector"])
    assert seen["clip"].shape == (1, 3, 224, 224)
    d = seen["enc"][0] - frame
    assert d.min() >= 0 and d.max() <= 0.02 + 1e-6                      # uniform [0, 1) * cond_aug, not Gaussian
    ref = torch.cat([O.timestep_embedding(torch.tensor([v]), 256) for v in (6.0, 127.0, 0.02)], -1)
    assert torch.allclose(c["vector"][0], ref[0], atol=1e-6)
    assert torch.allclose(sinusoid([127.0]), O.timestep_embedding(torch.tensor([127.0]), 256), atol=1e-6)


def test_conditioner_checkpoint_key_map():
    """conditioner.* keys of the reference checkpoint -> CLIP tower + cond-frame encoder (strict on the used sub-trees)."""
    import torch
    from oracle import cases
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig
    tv = cases.TINY_VAE
    ccfg, vcfg = ClipVisionConfig(width=320, layers=1, heads=4, image_size=56, embed_dim=1024), VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"])
    sd = {P.CKPT_PREFIXES["clip"] + k: v for k, v in init_by_name(OpenCLIPVisionTower(ccfg).spec(), seed=1).items()}
    sd.update({P.CKPT_PREFIXES["cond_encoder"] + k: v for k, v in init_by_name(CondFrameEncoder(vcfg).spec(), seed=2).items()})
    sd[P.CKPT_PREFIXES["clip"] + "logit_scale"] = torch.zeros(())                      # open_clip extras are ignored
    sd[P.CKPT_PREFIXES["cond_encoder"] + "post_quant_conv.weight"] = torch.zeros(4, 4, 1, 1)
    cond = P.load_conditioner(sd, device="cpu", clip_cfg=ccfg, vae_cfg=vcfg, num_frames=8)
    assert cond.T == 8 and cond.clip.device == "cpu"


def test_diffusers_vae_key_map_roundtrip():
    """diffusers AutoencoderKL key names -> sgm names: every key of our spec is produced exactly once with the right shape."""
    import torch
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencoderKL2D, VaeConfig, diffusers_vae_to_sgm_keys
    cfg = VaeConfig(32, (1, 2, 2, 4), 2)
    vae = AutoencoderKL2D(cfg)
    sgm = init_by_name(vae.spec(), seed=1)

    def to_diffusers(k, v):
        part, _, r = k.partition(".")
        if part in ("quant_conv", "post_quant_conv"):
            return k, v
        r = r.replace("norm_out.", "conv_norm_out.").replace("nin_shortcut.", "conv_shortcut.")
        r = r.replace("mid.block_1.", "mid_block.resnets.0.").replace("mid.block_2.", "mid_block.resnets.1.")
        if r.startswith("mid.attn_1."):
            r = r.replace("mid.attn_1.", "mid_block.attentions.0.").replace("proj_out.", "to_out.0.").replace("norm.", "group_norm.")
            for n in "qkv":
                r = r.replace(f"attentions.0.{n}.", f"attentions.0.to_{n}.")
            if r.endswith("weight") and v.dim() == 4:
                v = v[:, :, 0, 0]
        if r.startswith("down."):
            _, i, kind, rest = r.split(".", 3)
            r = f"down_blocks.{i}.resnets.{rest.split('.', 1)[0]}.{rest.split('.', 1)[1]}" if kind == "block" else f"down_blocks.{i}.downsamplers.0.{rest}"
        if r.startswith("up."):
            _, lvl, kind, rest = r.split(".", 3)
            i = 3 - int(lvl)
            r = f"up_blocks.{i}.resnets.{rest.split('.', 1)[0]}.{rest.split('.', 1)[1]}" if kind == "block" else f"up_blocks.{i}.upsamplers.0.{rest}"
        return part + "." + r, v

    dif = dict(to_diffusers(k, v) for k, v in sgm.items())
    assert "decoder.up_blocks.0.resnets.2.conv1.weight" in dif and "encoder.mid_block.attentions.0.to_q.weight" in dif
    assert dif["encoder.mid_block.attentions.0.to_q.weight"].dim() == 2
    back = diffusers_vae_to_sgm_keys(dif, 4)
    assert set(back) == set(sgm) and all(torch.equal(back[k], sgm[k]) for k in sgm)
    vae.load_state_dict(dif, device="cpu", diffusers_keys=True)


def test_enhance_codec_host_helpers():
    """frame-position planes (pipeline_i2vgen_xl.py:491-503) and the PIL centre crop (:965-1000) of the enhancer codec."""
    import numpy as np
    import PIL.Image
    import torch
    from streamingt2v_amd.enhance_codec import center_crop_wide, frame_position_planes
    lat = torch.randn(1, 4, 3, 5)
    x = frame_position_planes(lat, 5)
    assert x.shape == (1, 4, 5, 3, 5) and torch.equal(x[:, :, 0], lat)
    ref = []
    for frame_idx in range(4):                                   # the reference's loop
        ref.append(torch.ones_like(lat.unsqueeze(2)) * ((frame_idx + 1) / 4))
    assert torch.equal(x[:, :, 1:], torch.cat(ref, 2))
    assert frame_position_planes(lat, 1).shape == (1, 4, 1, 3, 5)
    img = PIL.Image.fromarray((np.random.rand(576, 1024, 3) * 255).astype("uint8"))
    assert center_crop_wide(img, (1280, 720)).size == (1280, 720) and center_crop_wide(img, (1280, 1280)).size == (1280, 1280)


def test_load_enhancer_from_diffusers_folder(tmp_path):
    """pipeline.load_enhancer reads a diffusers-format i2vgen-xl folder (config.json + safetensors per component, HF / diffusers key
    names) into the native UNet + codec; exercised with tiny random components written in that format."""
    import json
    import torch
    from safetensors.torch import save_file
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.clip_text import CLIPTextTower, ClipTextConfig
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencoderKL2D, VaeConfig

    def write(name, stem, cfg, sd):
        d = tmp_path / name
        d.mkdir()
        json.dump(cfg, open(d / "config.json", "w"))
        save_file({k: v.contiguous() for k, v in sd.items()}, str(d / f"{stem}.fp16.safetensors"))

    ucfg = dict(block_out_channels=[64, 128], layers_per_block=1, cross_attention_dim=128, down_block_types=["CrossAttnDownBlock3D", "DownBlock3D"])
    write("unet", "diffusion_pytorch_model", ucfg, init_by_name(I2VGenXLUNet(I2VConfig(block_out_channels=(64, 128), layers_per_block=1,
          cross_attention_dim=128, attn_levels=(True, False))).spec(), seed=1))
    # VAE written with DIFFUSERS key names (inverse of the map, as in test_diffusers_vae_key_map_roundtrip)
    sgm = init_by_name(AutoencoderKL2D(VaeConfig(32, (1, 2), 1)).spec(), seed=2)
    dif = {}
    for k, v in sgm.items():
        part, _, r = k.partition(".")
        if part in ("quant_conv", "post_quant_conv"):
            dif[k] = v; continue
        r = r.replace("norm_out.", "conv_norm_out.").replace("nin_shortcut.", "conv_shortcut.").replace("mid.block_1.", "mid_block.resnets.0.").replace("mid.block_2.", "mid_block.resnets.1.")
        if r.startswith("mid.attn_1."):
            r = r.replace("mid.attn_1.", "mid_block.attentions.0.").replace("proj_out.", "to_out.0.").replace("norm.", "group_norm.")
            for n in "qkv":
                r = r.replace(f"attentions.0.{n}.", f"attentions.0.to_{n}.")
            v = v[:, :, 0, 0] if v.dim() == 4 else v
        if r.startswith("down."):
            _, i, kind, rest = r.split(".", 3)
            r = f"down_blocks.{i}.resnets.{rest}" if kind == "block" else f"down_blocks.{i}.downsamplers.0.{rest}"
        if r.startswith("up."):
            _, lvl, kind, rest = r.split(".", 3)
            r = f"up_blocks.{1 - int(lvl)}.resnets.{rest}" if kind == "block" else f"up_blocks.{1 - int(lvl)}.upsamplers.0.{rest}"
        dif[part + "." + r] = v
    write("vae", "diffusion_pytorch_model", dict(block_out_channels=[32, 64], layers_per_block=1, scaling_factor=0.18215), dif)
    __import__("pytest").importorskip("transformers")
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    icfg = dict(hidden_size=320, intermediate_size=1280, num_hidden_layers=1, num_attention_heads=4, image_size=56, patch_size=14, projection_dim=128)
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_act="gelu", **icfg))
    write("image_encoder", "model", icfg, {k: v for k, v in hf.state_dict().items() if "position_ids" not in k})
    tcfg = dict(vocab_size=500, hidden_size=128, intermediate_size=512, num_hidden_layers=1, num_attention_heads=2, max_position_embeddings=77)
    write("text_encoder", "model", tcfg, init_by_name(CLIPTextTower(ClipTextConfig(**tcfg)).spec(), seed=3))
    unet, codec = P.load_enhancer(str(tmp_path), device="cpu")
    assert unet.cfg.block_out_channels == (64, 128) and codec.vae.sf == 0.18215 and codec.text_tower.cfg.layers == 1
    assert codec.image_tower.cfg.embed_dim == 128


def test_clip_tokenizer_matches_transformers_on_synthetic_vocab():
    """streamingt2v_amd.clip_tokenizer against transformers' CLIPTokenizer (the class the reference loads, pipeline_i2vgen_xl.py:213-231)
    on a synthetic byte-level BPE vocabulary (no real vocabulary offline): ids, framing, truncation, padding with either pad token."""
    pytest = __import__("pytest")
    pytest.importorskip("transformers"); pytest.importorskip("tokenizers")
    from collections import Counter
    from transformers import CLIPTokenizer
    from streamingt2v_amd.clip_tokenizer import CLIPBPETokenizer, bytes_to_unicode
    from streamingt2v_amd.pipeline import DEFAULTS
    corpus = (DEFAULTS["prompt"] + " " + DEFAULTS["negative_prompt"] + " a photo of an astronaut riding a horse on mars, it's 4k, don't blur; "
              "the quick brown fox jumps over the lazy dog 1234567890 times! high-quality detailed details quality").lower().split()
    be = bytes_to_unicode()
    words = Counter(tuple(list("".join(be[b] for b in w.encode())[:-1]) + ["".join(be[b] for b in w.encode())[-1] + "</w>"]) for w in corpus)
    merges = []
    for _ in range(120):                                               # plain BPE training: most frequent adjacent pair first
        pairs = Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _ = max(sorted(pairs.items()), key=lambda kv: kv[1])
        merges.append(f"{a} {b}")
        nw = Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and w[i] == a and w[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(w[i]); i += 1
            nw[tuple(out)] += c
        words = nw
    alphabet = list(be.values())
    tokens = alphabet + [c + "</w>" for c in alphabet] + [m.replace(" ", "") for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
    vocab = {t: i for i, t in enumerate(dict.fromkeys(tokens))}
    texts = [DEFAULTS["prompt"], DEFAULTS["negative_prompt"], "", "  A photo\tof an  astronaut, riding a horse!!  ", "it's don't we'll I'm they've he'd you're",
             "über café naïve 123 4k ... ?!", "quality " * 100, "<|endoftext|> tail", "emoji \U0001F600 and 中文 text"]
    for pad in ("<|endoftext|>", "!"):
        hf = CLIPTokenizer(vocab=vocab, merges=[tuple(m.split()) for m in merges], pad_token=pad, model_max_length=77)
        ours = CLIPBPETokenizer(vocab, merges, pad_token=pad)
        for t in texts:
            ref = hf(t, padding="max_length", max_length=77, truncation=True)
            got = ours(t)
            assert got["input_ids"] == ref["input_ids"], (pad, t, got["input_ids"][:20], ref["input_ids"][:20])
            assert got["attention_mask"] == ref["attention_mask"], (pad, t)
            assert len(got["input_ids"]) == 77
    both = ours([texts[0], texts[1]])
    assert both["input_ids"][1] == ours(texts[1])["input_ids"] and tuple(ours.input_ids(texts[0]).shape) == (1, 77)


def test_clip_tokenizer_from_pretrained_folder(tmp_path):
    import json
    from streamingt2v_amd.clip_tokenizer import CLIPBPETokenizer, bytes_to_unicode
    alphabet = list(bytes_to_unicode().values())
    tokens = alphabet + [c + "</w>" for c in alphabet] + ["hi</w>", "<|startoftext|>", "<|endoftext|>"]
    (tmp_path / "vocab.json").write_text(json.dumps({t: i for i, t in enumerate(tokens)}), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\nh i</w>\n", encoding="utf-8")
    (tmp_path / "special_tokens_map.json").write_text(json.dumps({"pad_token": "!", "bos_token": {"content": "<|startoftext|>"}}))
    (tmp_path / "tokenizer_config.json").write_text(json.dumps({"model_max_length": 16}))
    tok = CLIPBPETokenizer.from_pretrained(str(tmp_path))
    out = tok("Hi hi!")
    v = tok.encoder
    assert out["input_ids"] == [v["<|startoftext|>"], v["hi</w>"], v["hi</w>"], v["!"], v["<|endoftext|>"]] + [v["!"]] * 11
    assert out["attention_mask"] == [1] * 5 + [0] * 11


def test_kornia_resize_restatement_properties():
    """conditioner.kornia_resize_antialias (kornia 0.7.2 resize, antialias=True): no-op at equal size, plain interpolate when upscaling,
    constants preserved, and for 576x1024 -> 224x224 a (3, 7)-tap Gaussian with sigma ((576/224 - 1)/2, (1024/224 - 1)/2) before bicubic."""
    import torch.nn.functional as F
    from streamingt2v_amd.conditioner import _gaussian_kernel1d, kornia_resize_antialias
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, 36, 64, generator=g)
    assert kornia_resize_antialias(x, (36, 64)) is x
    assert torch.equal(kornia_resize_antialias(x, (72, 128)), F.interpolate(x, size=(72, 128), mode="bicubic", align_corners=True))
    c = torch.full((1, 3, 576, 1024), 0.37)
    assert (kornia_resize_antialias(c, (224, 224)) - 0.37).abs().max() < 1e-5
    big = torch.rand(1, 1, 576, 1024, generator=g)
    sy, sx = (576 / 224 - 1) / 2, (1024 / 224 - 1) / 2
    assert int(max(4 * sy, 3)) == 3 and int(max(4 * sx, 3)) == 7
    ky, kx = _gaussian_kernel1d(3, sy, "cpu"), _gaussian_kernel1d(7, sx, "cpu")
    k2 = ky[:, None] * kx[None, :]
    ref = F.conv2d(F.pad(big, (3, 3, 1, 1), mode="reflect"), k2[None, None])
    ref = F.interpolate(ref, size=(224, 224), mode="bicubic", align_corners=True)
    assert (kornia_resize_antialias(big, (224, 224)) - ref).abs().max() < 1e-5


def test_autoregressive_loop_matches_reference_golden():
    """streaming_svd.StreamingSVD._autoregressive_generation (product host logic) against the uint8 video of the reference's UNMODIFIED loop
    (tests/golden/ar_loop_tiny.pt, oracle/make_golden_ar_loop.py): same stand-in chunk generator on both sides, so anchor / control-frame
    hand-over, kept frames, concatenation and the final truncation (range oracle, bit-exact with the reference's IImage path) must agree."""
    from oracle.cases import TINY_AR, tiny_ar_chunk0, tiny_ar_generate
    from oracle.range_oracle import frames_to_uint8
    import os
    from streamingt2v_amd.streaming_svd import StreamingSVD
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ar_loop_tiny.pt"))
    calls = []

    class Host(StreamingSVD):
        def __init__(self):
            self.num_conditional_frames = TINY_AR["Tc"]

        def _generate_conditional_output(self, c, uc, ctrl_frames, noise, num_steps=None):
            calls.append((c.clone(), ctrl_frames.clone()))
            return tiny_ar_generate(c, ctrl_frames, len(calls) - 1)

    video = Host()._autoregressive_generation(tiny_ar_chunk0(), lambda anchor: (anchor, None), TINY_AR["n_ar"], [None] * TINY_AR["n_ar"],
                                              anchor_index=TINY_AR["anchor"])
    for k in range(TINY_AR["n_ar"]):
        assert torch.equal(calls[k][0], g["anchors"][k]) and torch.equal(calls[k][1], g["ctrl"][k]), k
    assert torch.equal(frames_to_uint8(video), g["video"])


def test_chunk0_goes_through_uint8_like_the_reference():
    """image_to_video :388-394: chunk 0 = ToTensor(PIL uint8 frame) * 2 - 1; every value sits on the 1/255 grid and survives a second pass."""
    from oracle.cases import tiny_ar_chunk0
    from streamingt2v_amd.streaming_svd import StreamingSVD
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4, 3, 8, 8, generator=g) * 2.4 - 1.2
    q = StreamingSVD.quantize_like_pil(x)
    u8 = ((x / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8)                      # diffusers numpy_to_pil
    assert torch.equal(q, u8.float() / 255.0 * 2.0 - 1)                                  # torchvision ToTensor, then * 2 - 1
    assert torch.equal(StreamingSVD.quantize_like_pil(q), q) and torch.equal(StreamingSVD.quantize_like_pil(tiny_ar_chunk0()), tiny_ar_chunk0())


def test_enhance_video_structure_matches_reference_process(monkeypatch):
    """pipeline.StreamingPipeline.enhance_video against the recorded calls of the reference's UNMODIFIED i2v_enhance_process
    (tests/golden/enhance_process_tiny.pt, oracle/make_golden_enhance_process.py): key frames, the key-frame pre-pass whose decoded output
    becomes the per-window images, truncation to whole windows, chunk / overlap / window length, strength, 30 steps, guidance 9."""
    import os
    import zlib
    import numpy as np
    from oracle.cases import tiny_enhance_process_inputs
    from streamingt2v_amd import enhance, pipeline
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "enhance_process_tiny.pt"))
    crc = lambda f: zlib.crc32(np.ascontiguousarray(np.asarray(f)).tobytes())
    log = []

    class Codec:
        w, h = 12, 8                                  # the stand-in frames' size: the front-end key-image resize is a no-op here

        def window_conditioning(self, images, n, window_len):
            log.append(dict(image=[crc(i) for i in images], num_frames=window_len, n_windows=n))
            return [dict() for _ in range(n)]

        def encode_video(self, frames):
            log[-1]["video"] = [crc(f) for f in frames]
            return list(frames)

        def noise_like(self, lat):
            return None

        def decode(self, lat):
            return [255 - np.asarray(f) for f in lat]

    class Enhancer:
        def __init__(self, unet, scheduler=None, guidance_scale=None, num_inference_steps=None, strength=None):
            self.kw = dict(guidance_scale=guidance_scale, num_inference_steps=num_inference_steps, strength=strength)

        def denoise(self, lat, noise, conds, chunk_size, overlap_size, rng=None, group=None):
            log[-1].update(self.kw, chunk_size=chunk_size, overlap_size=overlap_size, n_conds=len(conds))
            return lat

    monkeypatch.setattr(enhance, "I2VEnhancer", Enhancer)
    pipe = object.__new__(pipeline.StreamingPipeline)
    pipe.cfg, pipe.enhancer_unet, pipe.enhance_codec = dict(pipeline.DEFAULTS), object(), Codec()
    for name, n_frames, rb, chunk, overlap in (("blend_100", 100, True, 38, 12), ("blend_90", 90, True, 38, 12), ("plain_100", 100, False, 100, 0)):
        image, video = tiny_enhance_process_inputs(n_frames)
        log.clear()
        out = pipe.enhance_video(image[0], video, chunk_size=chunk, overlap_size=overlap, strength=0.97, use_randomized_blending=rb)
        ref = gold[name]
        assert len(log) == len(ref["calls"]), name
        for got, want in zip(log, ref["calls"]):
            for k in ("image", "video", "chunk_size", "overlap_size", "num_frames", "strength", "num_inference_steps", "guidance_scale"):
                assert got[k] == want[k], (name, k)
            assert got["n_conds"] == got["n_windows"] == len(want["image"])
            d = pipeline.DEFAULTS            # what the reference passes besides (i2v_enhance_interface.py:87-88, 99-133)
            assert (want["prompt"], want["negative_prompt"]) == (d["prompt"], d["negative_prompt"])
            assert (want["height"], want["width"], want["decode_chunk_size"]) == (d["enhance_height"], d["enhance_width"], 1)
            assert "target_fps" not in want and d["enhance_target_fps"] == 38          # not passed: the fork's __call__ default applies
        assert [crc(f) for f in out] == ref["out"], name


def test_front_end_image_handling_matches_reference_enhance_video():
    """What inference_i2v.StreamingPipeline.enhance_video (unmodified, tests/golden/frontend_enhance_tiny.pt) hands the enhancement pipeline:
    key image via IImage.resize (BICUBIC to 1280 x 720), frames via PIL's default resize -- vs pipeline.resize_key_image and
    EnhanceCodec.encode_video's frame resize (observed through a recording VAE stand-in)."""
    import os
    import numpy as np
    from oracle.cases import tiny_frontend_enhance_inputs
    from streamingt2v_amd.enhance_codec import EnhanceCodec
    from streamingt2v_amd.pipeline import resize_key_image
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_enhance_tiny.pt"))
    image, video = tiny_frontend_enhance_inputs()
    key = resize_key_image(image)
    assert key.size == (1280, 720) and np.array_equal(np.asarray(key)[::40, ::40], g["image"][0].numpy())
    assert resize_key_image(key) is key
    seen = []

    class Vae:
        def encode_sample(self, x, generator=None):
            seen.append(x.clone())
            return torch.zeros(x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8)

    EnhanceCodec(Vae(), None, None, device="cpu").encode_video(list(video))
    px = torch.cat(seen, 0)                                                           # [F, 3, 720, 1280] in [-1, 1]
    back = ((px + 1.0) * 127.5).round().to(torch.uint8).permute(0, 2, 3, 1).numpy()
    for i in range(len(video)):
        assert np.array_equal(back[i][::40, ::40], g["video"][i].numpy()), i


def test_svd_conditioner_matches_reference_conditioning_path(monkeypatch):
    """conditioner.SVDConditioner against what the reference's UNMODIFIED _generate_conditional_output + GeneralConditioner + get_batch_sgm
    hand the sampler (tests/golden/svd_conditioning_tiny.pt, oracle/make_golden_conditioner.py; OpenCLIP tower and VAE encoder replaced by the
    same linear stand-ins on both sides, the CLIP preprocessing -- which lives inside the replaced tower -- switched off)."""
    import os
    from oracle.cases import TINY_SVD_COND as cfg, fake_clip_embed, fake_cond_encode, tiny_svd_cond_inputs
    from streamingt2v_amd.conditioner import SVDConditioner
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "svd_conditioning_tiny.pt"))
    monkeypatch.setattr(SVDConditioner, "clip_preprocess", staticmethod(lambda img: img))
    cond = SVDConditioner(fake_clip_embed, fake_cond_encode, num_frames=cfg["T"], generator=None)
    torch.manual_seed(cfg["seed"])                                     # cond_frames = image + 0.02 * rand_like(image): global stream (:174)
    c, uc = cond(tiny_svd_cond_inputs()["frame"])
    for d, gd in ((c, g["c"]), (uc, g["uc"])):
        for k in ("crossattn", "concat", "vector"):
            assert d[k].shape[0] == cfg["T"] and tuple(d[k].shape[1:]) == tuple(gd[k].shape[1:]), k
            assert all(torch.equal(d[k][0], d[k][i]) for i in range(1, cfg["T"]))
            assert (d[k][:2] - gd[k]).abs().max() <= 1e-6, (k, (d[k][:2] - gd[k]).abs().max())
    assert g["randn_shape"] == (cfg["T"], 4, cfg["H"] // 8, cfg["W"] // 8) and g["ctrl_equal"]
    assert g["extra"]["batch_size"] == 2 and g["extra"]["num_conditional_frames"] == cfg["Tc"] and g["extra"]["num_video_frames"] == cfg["T"]
    assert tuple(g["extra"]["image_only_indicator"].shape) == (2, cfg["T"]) and not g["extra"]["image_only_indicator"].any()


def test_resize_and_keep_matches_pil_semantics():
    """pipeline.resize_and_keep = utils/inference_utils.py:36-41: PIL default (BICUBIC) resize to 576 rows, width int(w * 576 / h)."""
    import numpy as np
    import PIL.Image
    from streamingt2v_amd.pipeline import resize_and_keep
    rs = np.random.default_rng(3)
    for (h, w) in ((1080, 1920), (576, 1024), (300, 533), (720, 1281)):
        a = rs.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out = resize_and_keep(a)
        wsize = int(float(w) * (576 / float(h)))
        assert out.shape == (576, wsize, 3) and np.array_equal(out, np.asarray(PIL.Image.fromarray(a).resize((wsize, 576))))


def test_from_pretrained_assembles_stage1_conditioner_and_vfi():
    """StreamingPipeline.from_pretrained on a tiny checkpoint with the reference's key prefixes (UNet, ControlNet, decoder, OpenCLIP tower,
    cond-frame encoder) plus an EMA-VFI state_dict: every stage object is built and wired (CPU: construction only)."""
    from oracle import cases
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from streamingt2v_amd.conditioner import SVDConditioner
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    tu, tv = cases.TINY_UNET, cases.TINY_VAE
    ucfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"],
                      channel_mult=tu["channel_mult"], conditioning_embedding_out_channels=tu["cond_embed"])
    vcfg, ccfg, ecfg = VaeConfig(tv["ch"], tv["ch_mult"], tv["num_res_blocks"]), ClipVisionConfig(width=320, layers=1, heads=4, embed_dim=1024), VaeConfig(32, (1, 1, 1, 2), 1)
    fcfg = VFIConfig(F=cases.TINY_VFI["F"], depth=cases.TINY_VFI["depth"])
    sd = {}
    for prefix, mod, seed in ((P.CKPT_PREFIXES["unet"], VideoUNet(ucfg), 1), (P.CKPT_PREFIXES["controlnet"], ControlNet(ucfg), 2),
                              (P.CKPT_PREFIXES["decoder"], VideoDecoder(vcfg), 3), (P.CKPT_PREFIXES["clip"], OpenCLIPVisionTower(ccfg), 4),
                              (P.CKPT_PREFIXES["cond_encoder"], CondFrameEncoder(ecfg), 5)):
        sd.update({prefix + k: v for k, v in init_by_name(mod.spec(), seed=seed).items()})
    pipe = P.StreamingPipeline.from_pretrained(sd, vfi_ckpt=cases.vfi_weights(EMAVFI(fcfg).spec()), device="cpu", unet_cfg=ucfg, vae_cfg=vcfg,
                                               clip_cfg=ccfg, cond_vae_cfg=ecfg, vfi_cfg=fcfg, num_frames_per_chunk=tu["T"], num_conditional_frames=tu["Tc"])
    assert isinstance(pipe.conditioner, SVDConditioner) and pipe.conditioner.T == tu["T"] and isinstance(pipe.vfi, EMAVFI) and pipe.vfi.loaded
    assert pipe.cfg["input_height"] == 576 and pipe.enhancer_unet is None


def test_defaults_match_reference_config_values():
    """The product's hard-coded defaults against the values read from the reference's config.yaml (tests/golden/config_values.json,
    oracle/make_golden_config.py)."""
    import inspect
    import json
    import os
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.sampling import AlignYourSteps, EulerEDMSampler, VScalingWithEDMcNoise
    from streamingt2v_amd.streaming_svd import StreamingSVD
    from streamingt2v_amd.temporal_ae import VaeConfig
    from streamingt2v_amd.video_model import UNetConfig
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "config_values.json")) as f:
        g = json.load(f)
    u, n = UNetConfig(), g["network"]
    for k in ("in_channels", "model_channels", "out_channels", "num_res_blocks", "context_dim", "adm_in_channels", "num_head_channels", "controlnet_mode"):
        assert getattr(u, k) == n[k], k
    assert list(u.attention_resolutions) == n["attention_resolutions"] and list(u.channel_mult) == n["channel_mult"]
    assert list(u.conditioning_embedding_out_channels) == g["controlnet"]["conditioning_embedding_out_channels"]
    assert n["use_apm"] is False and n["merging_mode"] == "attention_cross_attention" and n["transformer_depth"] == 1      # what the kernels assume
    s = EulerEDMSampler()
    sm = g["sampler"]
    assert s.num_steps == sm["num_steps"] and isinstance(s.discretization, AlignYourSteps) and s.discretization.sigma_max == sm["sigma_max"]
    assert s.guider.num_frames == sm["num_frames"] and float(s.guider.scale[0]) == sm["min_scale"] and float(s.guider.scale[-1]) == sm["max_scale"]
    assert (sm["s_churn"], sm["s_tmin"], sm["s_noise"]) == (0.0, 0.0, 1.0)            # the plain Euler step the product implements
    assert isinstance(s.scaling, VScalingWithEDMcNoise) and g["denoiser_scaling"] == "VScalingWithEDMcNoise"
    d, v = P.DEFAULTS, VaeConfig()
    assert d["seed"] == g["seed_everything"] and d["num_steps"] == sm["num_steps"] and d["num_frames_per_chunk"] == sm["num_frames"]
    assert d["num_conditional_frames"] == g["inference"]["num_conditional_frames"]
    sig = inspect.signature(StreamingSVD._autoregressive_generation)
    assert sig.parameters["anchor_index"].default == int(g["inference"]["anchor_frames"])
    assert inspect.signature(StreamingSVD.__init__).parameters["scale_factor"].default == g["scale_factor"]
    assert (v.ch, list(v.ch_mult), v.num_res_blocks, v.z_channels, v.out_ch) == tuple(g["decoder"][k] for k in ("ch", "ch_mult", "num_res_blocks", "z_channels", "out_ch"))


def test_ddim_schedule_from_config_and_spacings():
    """enhance.DDIMSchedule.from_config (scheduler/scheduler_config.json of the checkpoint folder): the i2vgen-xl values reproduce the default
    schedule; the three diffusers timestep spacings; update-rule options that are not implemented are refused."""
    import pytest
    from streamingt2v_amd.enhance import DDIMSchedule
    i2v = dict(_class_name="DDIMScheduler", beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, num_train_timesteps=1000,
               prediction_type="v_prediction", rescale_betas_zero_snr=True, set_alpha_to_one=False, steps_offset=1, timestep_spacing="leading", thresholding=False)
    a, b = DDIMSchedule.from_config(i2v), DDIMSchedule()
    assert torch.equal(a.alphas_cumprod, b.alphas_cumprod) and a.set_timesteps(30) == b.set_timesteps(30) and a.alphas(201) == b.alphas(201)
    assert b.set_timesteps(10) == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]
    assert DDIMSchedule.from_config(dict(i2v, timestep_spacing="trailing", steps_offset=0)).set_timesteps(10) == [999, 899, 799, 699, 599, 499, 399, 299, 199, 99]
    assert DDIMSchedule.from_config(dict(i2v, timestep_spacing="linspace", steps_offset=0)).set_timesteps(4) == [999, 666, 333, 0]
    lin = DDIMSchedule.from_config(dict(beta_schedule="linear", beta_start=0.0001, beta_end=0.02))
    assert not lin.v_prediction and abs(float(lin.alphas_cumprod[-1]) - 4.0358e-05) < 1e-7 and lin.final_alpha_cumprod == 1.0
    for bad in (dict(i2v, clip_sample=True), dict(i2v, thresholding=True), dict(i2v, beta_schedule="squaredcos_cap_v2"), dict(i2v, prediction_type="sample")):
        with pytest.raises(NotImplementedError):
            DDIMSchedule.from_config(bad)


def test_stock_svd_xt_key_maps_and_folder_loader(tmp_path):
    """Chunk 0 runs on the STOCK SVD-XT weights in the reference (config.yaml:280-299; streaming_svd.py:388-390): diffusers_keys.py maps the
    diffusers names of UNetSpatioTemporalConditionModel / AutoencoderKLTemporalDecoder onto the sgm-named specs.  Pinned here: the map is a
    bijection on the shipped architecture, the totals are the published ones, names every diffusers checkpoint of this model carries come out,
    and pipeline.load_stock_svd_xt reads a folder written in that format (tiny components), strictly."""
    import json
    import pytest
    from safetensors.torch import save_file
    from streamingt2v_amd import pipeline as P
    from streamingt2v_amd.diffusers_keys import sgm_temporal_decoder_key_to_diffusers as dkey, sgm_unet_key_to_diffusers as ukey
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import CondFrameEncoder, VaeConfig, VideoDecoder
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    # ---- shipped architecture: names only
    spec = VideoUNet(UNetConfig(controlnet_mode=False)).spec()
    names = [ukey(n) for n in spec.names()]
    assert len(set(names)) == len(names) == 1428 and spec.numel() == 1_524_623_082          # SVD-XT UNet: 1.52 B parameters
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias", "conv_norm_out.weight", "conv_out.bias",
              "down_blocks.0.resnets.0.spatial_res_block.conv1.weight", "down_blocks.0.resnets.0.temporal_res_block.time_emb_proj.weight",
              "down_blocks.0.resnets.1.time_mixer.mix_factor", "down_blocks.1.resnets.0.spatial_res_block.conv_shortcut.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight", "down_blocks.2.attentions.1.temporal_transformer_blocks.0.ff_in.net.0.proj.weight",
              "down_blocks.0.attentions.1.time_pos_embed.linear_1.weight", "down_blocks.0.attentions.0.time_mixer.mix_factor", "down_blocks.2.downsamplers.0.conv.weight",
              "down_blocks.3.resnets.1.temporal_res_block.conv2.bias", "mid_block.resnets.1.spatial_res_block.norm2.weight", "mid_block.attentions.0.proj_in.weight",
              "up_blocks.0.resnets.2.spatial_res_block.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.1.attentions.2.temporal_transformer_blocks.0.attn1.to_out.0.bias",
              "up_blocks.2.upsamplers.0.conv.bias", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2.weight", "up_blocks.3.resnets.2.time_mixer.mix_factor"):
        assert k in names, k
    assert not any(n.startswith(("down_blocks.3.attentions", "down_blocks.3.downsamplers", "up_blocks.0.attentions", "up_blocks.3.upsamplers")) for n in names)
    dspec, espec = VideoDecoder().spec(), CondFrameEncoder(VaeConfig()).spec()
    dnames = [dkey(n) for n in dspec.names()]
    assert len(set(dnames)) == len(dnames) and dspec.numel() + espec.numel() == 97_742_847      # AutoencoderKLTemporalDecoder: 97.7 M
    for k in ("decoder.conv_in.weight", "decoder.time_conv_out.weight", "decoder.conv_out.bias", "decoder.conv_norm_out.weight", "decoder.mid_block.attentions.0.to_q.weight",
              "decoder.mid_block.attentions.0.group_norm.bias", "decoder.mid_block.resnets.1.temporal_res_block.conv1.weight", "decoder.up_blocks.0.resnets.2.time_mixer.mix_factor",
              "decoder.up_blocks.2.resnets.0.spatial_res_block.conv_shortcut.weight", "decoder.up_blocks.2.upsamplers.0.conv.weight", "decoder.up_blocks.3.resnets.1.spatial_res_block.norm1.weight"):
        assert k in dnames, k
    assert not any(n.startswith("decoder.up_blocks.3.upsamplers") for n in dnames)
    # ---- tiny components written as a diffusers folder
    def write(name, stem, cfg, sd):
        d = tmp_path / name
        d.mkdir()
        json.dump(cfg, open(d / "config.json", "w"))
        save_file({k: v.contiguous() for k, v in sd.items()}, str(d / f"{stem}.fp16.safetensors"))

    ucfg = UNetConfig(model_channels=64, num_res_blocks=1, attention_resolutions=(1,), channel_mult=(1, 2), context_dim=128, controlnet_mode=False)
    usd = init_by_name(VideoUNet(ucfg).spec(), seed=11)
    write("unet", "diffusion_pytorch_model", dict(block_out_channels=[64, 128], layers_per_block=1, cross_attention_dim=128, in_channels=8, out_channels=4,
                                                  projection_class_embeddings_input_dim=768, num_frames=5,
                                                  down_block_types=["CrossAttnDownBlockSpatioTemporal", "DownBlockSpatioTemporal"]), {ukey(k, 1): v for k, v in usd.items()})
    vcfg = VaeConfig(32, (1, 2), 1)
    dsd, esd = init_by_name(VideoDecoder(vcfg).spec(), seed=12), init_by_name(CondFrameEncoder(vcfg).spec(), seed=13)
    dif = {}
    for k, v in dsd.items():
        nk = dkey(k, 2)
        dif[nk] = v[:, :, 0, 0] if "attentions.0.to_" in nk and v.dim() == 4 else v
    for k, v in esd.items():                                  # the encoder half in diffusers' names (inverse of temporal_ae.diffusers_vae_to_sgm_keys)
        part, _, r = k.partition(".")
        if part == "quant_conv":
            dif[k] = v; continue
        r = r.replace("norm_out.", "conv_norm_out.").replace("nin_shortcut.", "conv_shortcut.").replace("mid.block_1.", "mid_block.resnets.0.").replace("mid.block_2.", "mid_block.resnets.1.")
        if r.startswith("mid.attn_1."):
            r = r.replace("mid.attn_1.", "mid_block.attentions.0.").replace("proj_out.", "to_out.0.").replace("norm.", "group_norm.")
            for n in "qkv":
                r = r.replace(f"attentions.0.{n}.", f"attentions.0.to_{n}.")
            v = v[:, :, 0, 0] if v.dim() == 4 else v
        if r.startswith("down."):
            _, i, kind, rest = r.split(".", 3)
            r = f"down_blocks.{i}.resnets.{rest}" if kind == "block" else f"down_blocks.{i}.downsamplers.0.{rest}"
        dif["encoder." + r] = v
    write("vae", "diffusion_pytorch_model", dict(block_out_channels=[32, 64], layers_per_block=1), dif)
    pytest.importorskip("transformers")
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    icfg = dict(hidden_size=320, intermediate_size=1280, num_hidden_layers=1, num_attention_heads=4, image_size=56, patch_size=14, projection_dim=128)
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_act="gelu", **icfg))
    write("image_encoder", "model", icfg, {k: v for k, v in hf.state_dict().items() if "position_ids" not in k})
    wrap, fsm, cond = P.load_stock_svd_xt(str(tmp_path), device="cpu", num_conditional_frames=2)
    assert isinstance(wrap, StreamingWrapper) and wrap.controlnet is None and not wrap.diffusion_model.controlnet_mode and cond.T == 5
    assert wrap.diffusion_model.cfg.channel_mult == (1, 2) and wrap.diffusion_model.cfg.attention_resolutions == (1,) and hasattr(cond, "first_chunk")
    # strictness: a tensor the architecture does not have, and a missing one
    import os
    from safetensors.torch import load_file
    f = str(tmp_path / "unet" / "diffusion_pytorch_model.fp16.safetensors")
    sd = load_file(f)
    save_file(dict(sd, **{"down_blocks.1.attentions.0.proj_in.weight": torch.zeros(1)}), f)
    with pytest.raises(KeyError, match="without a counterpart"):
        P.load_stock_svd_xt(str(tmp_path), device="cpu")
    sd.pop("mid_block.resnets.0.time_mixer.mix_factor")
    save_file(sd, f)
    with pytest.raises(KeyError, match="no tensor"):
        P.load_stock_svd_xt(str(tmp_path), device="cpu")


def test_rowgemm320_weight_image_layout_and_round_trip():
    """video_model.pack_rowgemm320 (the image svd_rowgemm320 reads, include/svdhip.h): fragment 10 s + o, lane l, element e = W[32 o + l % 32][16 s + 8 (l // 32) + e];
    tests/svd_shim._rowgemm_unpack (the CPU statement of the kernel's view of it) inverts it exactly."""
    import torch
    from streamingt2v_amd import ops
    from streamingt2v_amd.video_model import pack_rowgemm320
    from tests import svd_shim
    prev = ops.ELEM
    ops.ELEM = torch.float32
    try:
        w = torch.arange(320 * 320, dtype=torch.float32).reshape(320, 320)          # every element distinct: W[n][k] = 320 n + k
        img = pack_rowgemm320(w)
        assert img.dtype == torch.uint8 and img.numel() == 200 * 64 * 8 * 4
        f = img.view(torch.float32).view(20, 10, 64, 8)
        for s_, o, l, e in ((0, 0, 0, 0), (3, 7, 37, 5), (19, 9, 63, 7), (5, 0, 31, 0), (4, 9, 32, 3)):
            assert f[s_, o, l, e].item() == 320 * (32 * o + l % 32) + 16 * s_ + 8 * (l // 32) + e
        assert torch.equal(svd_shim._rowgemm_unpack(img, torch.float32), w)
    finally:
        ops.ELEM = prev


def test_rowproj320_weight_image_layout_and_round_trip():
    """video_model.pack_rowproj320 (the image svd_rowproj320 reads, include/svdhip.h): chunk ch, fragment 2 s + t, lane l, element e =
    W[64 ch + 2 (l % 32) + t][16 s + 8 (l // 32) + e] -- the chunk's channels interleaved over the two column tiles; tests/svd_shim._rowproj_unpack inverts it."""
    import torch
    from streamingt2v_amd import ops
    from streamingt2v_amd.video_model import pack_rowproj320
    from tests import svd_shim
    prev = ops.ELEM
    ops.ELEM = torch.float32
    try:
        N = 960
        w = torch.arange(N * 320, dtype=torch.float32).reshape(N, 320)               # every element distinct: W[n][k] = 320 n + k
        img = pack_rowproj320(w)
        assert img.dtype == torch.uint8 and img.numel() == (N // 64) * 40 * 64 * 8 * 4
        f = img.view(torch.float32).view(N // 64, 20, 2, 64, 8)
        for ch, s_, t, l, e in ((0, 0, 0, 0, 0), (14, 19, 1, 63, 7), (3, 7, 1, 37, 5), (9, 4, 0, 32, 3), (1, 0, 1, 31, 0)):
            assert f[ch, s_, t, l, e].item() == 320 * (64 * ch + 2 * (l % 32) + t) + 16 * s_ + 8 * (l // 32) + e
        assert torch.equal(svd_shim._rowproj_unpack(img, torch.float32, N), w)
    finally:
        ops.ELEM = prev
