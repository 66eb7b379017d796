"""Seeded test cases shared by oracle/make_golden.py (reference side) and tests/ (oracle + HIP side).

TEST INFRASTRUCTURE.  Inputs are re-derived from fixed seeds so that only reference OUTPUTS need to be stored
under tests/golden/.  Sizes: the smallest configuration that still exercises every kernel path -- the ControlNet
hard-codes model_channels = 320 (controlnet.py:443-446), so "tiny" shrinks depth/levels/frames/pixels instead.
"""
import torch

TINY_UNET = dict(num_res_blocks=1, attention_resolutions=(2, 1), channel_mult=(1, 2), cond_embed=(32, 96, 256, 512),
                 T=8, Tc=3, h=16, w=16)
TINY_VAE = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1, T=4, h=8, w=8)


def _common_unet_kwargs():
    return dict(in_channels=8, model_channels=320, out_channels=4, num_conditional_frames=None, dropout=0.0,
                conv_resample=True, dims=2, num_classes="sequential", use_checkpoint=False, num_heads=-1,
                num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                transformer_depth=1, transformer_depth_middle=None, context_dim=1024, time_downup=False,
                time_context_dim=None, extra_ff_mix_layer=True, use_spatial_context=True,
                merge_strategy="learned_with_images", merge_factor=0.5, spatial_transformer_attn_type="softmax",
                video_kernel_size=[3, 1, 1], use_linear_in_transformer=True, adm_in_channels=768,
                disable_temporal_crossattention=False, max_ddpm_temb_period=10000,
                merging_mode="attention_cross_attention", controlnet_mode=True, use_apm=False)


def tiny_unet_kwargs():
    """Constructor kwargs of the reference VideoUNet (config.yaml:69-115) shrunk to the tiny case."""
    kw = _common_unet_kwargs()
    kw.update(num_res_blocks=TINY_UNET["num_res_blocks"], attention_resolutions=list(TINY_UNET["attention_resolutions"]),
              channel_mult=list(TINY_UNET["channel_mult"]))
    return kw


def full_unet_kwargs():
    kw = _common_unet_kwargs()
    kw.update(num_res_blocks=2, attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4])
    return kw


def _gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def tiny_wrapper_inputs():
    g = _gen(1234)
    T, h, w = TINY_UNET["T"], TINY_UNET["h"], TINY_UNET["w"]
    F = 2 * T
    return dict(
        x=torch.randn(F, 4, h, w, generator=g),
        t=torch.randn(F, generator=g) * 0.5,
        concat=torch.randn(F, 4, h, w, generator=g) * 0.5,
        crossattn=torch.randn(F, 1, 1024, generator=g),
        vector=torch.randn(F, 768, generator=g) * 0.5,
        ctrl_frames=torch.rand(1, TINY_UNET["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1,
    )


def tiny_sampler_inputs():
    g = _gen(4321)
    T, h, w = TINY_UNET["T"], TINY_UNET["h"], TINY_UNET["w"]
    c = dict(concat=torch.randn(T, 4, h, w, generator=g) * 0.5, crossattn=torch.randn(T, 1, 1024, generator=g),
             vector=torch.randn(T, 768, generator=g) * 0.5)
    uc = dict(concat=torch.zeros(T, 4, h, w), crossattn=torch.zeros(T, 1, 1024), vector=c["vector"].clone())
    return dict(noise=torch.randn(T, 4, h, w, generator=g), c=c, uc=uc)


def tiny_vae_inputs():
    g = _gen(99)
    z = torch.randn(TINY_VAE["T"], 4, TINY_VAE["h"], TINY_VAE["w"], generator=g)
    # encoder input: 2 frames of 64x128 pixels in [-1, 1]  (2 levels -> latent 32x64: 2048 tokens for the mid attention)
    return dict(z=z, x_enc=torch.rand(2, 3, 64, 128, generator=g) * 2 - 1)


# ---- I2VGen-XL enhancer (row A12): tiny configuration of code/i2v_enhance/unet_i2vgen_xl.py:188-211 ----
# 3 levels (cross-attn, cross-attn, plain), one layer per block, odd latent height so that the up path has to forward the
# upsample size (the shipped 90x160 latent does too: 90 % 8 != 0).
TINY_I2V = dict(block_out_channels=(64, 128, 128), down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"), layers_per_block=1,
                norm_num_groups=32, cross_attention_dim=128, attention_head_dim=64, F=6, h=9, w=16, text_tokens=7)


def tiny_i2v_kwargs():
    return {k: TINY_I2V[k] for k in ("block_out_channels", "down_block_types", "up_block_types", "layers_per_block",
                                     "norm_num_groups", "cross_attention_dim", "attention_head_dim")}


def tiny_i2v_inputs():
    g = _gen(777)
    B, Fr, h, w, cd = 2, TINY_I2V["F"], TINY_I2V["h"], TINY_I2V["w"], TINY_I2V["cross_attention_dim"]
    return dict(sample=torch.randn(B, 4, Fr, h, w, generator=g), t=torch.tensor(481), fps=torch.tensor([8, 8]),
                image_latents=torch.randn(B, 4, Fr, h, w, generator=g) * 0.7,
                image_embeddings=torch.randn(B, cd, generator=g), text=torch.randn(B, TINY_I2V["text_tokens"], cd, generator=g))


def range_inputs():
    """Row A13: frames [4, 3, 16, 32] in [-1, 1]: seeded uniform values, plus next to every integer boundary k / 127.5 - 1 the
    fp32 neighbours (-2 .. +2 ulp) where the reference's truncating uint8 conversion flips."""
    g = _gen(4242)
    x = torch.rand(4, 3, 16, 32, generator=g) * 2 - 1
    k = torch.arange(0, 256, dtype=torch.float64)
    edge = (k / 127.5 - 1).float()
    flat = x.view(-1)
    vals = []
    for d in (-2, -1, 0, 1, 2):
        e = edge.clone()
        for _ in range(abs(d)):
            e = torch.nextafter(e, torch.full_like(e, 2.0 if d > 0 else -2.0))
        vals.append(e)
    vals = torch.cat(vals).clamp(-1, 1)
    flat[: vals.numel()] = vals
    return x


# ---- EMA-VFI (code/i2v_enhance/thirdparty/VFI): tiny configuration F=8 (production F=32), same depths / window / head dim ----
TINY_VFI = dict(F=8, depth=(2, 2, 2, 4, 4), H=48, W=80)


def vfi_weights(spec, seed=11):
    """init_by_name, except PReLU slopes (1-D ``.weight`` of a non-norm layer) ~ 0.25 + 0.1 N so that the negative branch matters."""
    from streamingt2v_amd.params import init_by_name
    sd = init_by_name(spec, seed=seed)
    for name, shape in spec:
        if len(shape) == 1 and name.endswith(".weight") and "norm" not in name:
            sd[name] = 0.25 + 0.1 * (sd[name] - 1.0) / 0.1
    return sd


def tiny_vfi_inputs():
    """Two smooth-ish frames in [0, 1] (BGR order is irrelevant here): low-resolution noise upsampled, second frame shifted."""
    g = _gen(2024)
    H, W = TINY_VFI["H"], TINY_VFI["W"]
    base = torch.nn.functional.interpolate(torch.rand(1, 3, H // 4 + 2, W // 4 + 2, generator=g), scale_factor=4, mode="bicubic",
                                           align_corners=False).clamp(0, 1)
    return dict(img0=base[:, :, 2:2 + H, 3:3 + W].contiguous(), img1=base[:, :, 4:4 + H, 1:1 + W].contiguous())


# ---- the enhancer's pipeline call (code/i2v_enhance/pipeline_i2vgen_xl.py:607-935) around the tiny UNet: 10 frames of 72x128 pixels,
#      two blending windows of 6 frames with overlap 2, key images of other sizes / aspect ratios, 10 DDIM steps at strength 0.35 (3 run) ----
TINY_I2V_CALL = dict(H=72, W=128, n_frames=10, chunk=6, overlap=2, steps=10, strength=0.35, guidance=9.0, fps=38, gen_seed=8888, py_seed=33)


def tiny_i2v_call_inputs():
    import numpy as np
    import PIL.Image
    c = TINY_I2V_CALL
    rs = np.random.default_rng(4)
    frames = [rs.integers(0, 256, (c["H"], c["W"], 3), dtype=np.uint8) for _ in range(c["n_frames"])]
    images = [PIL.Image.fromarray(rs.integers(0, 256, (150, 200, 3), dtype=np.uint8)), PIL.Image.fromarray(rs.integers(0, 256, (90, 300, 3), dtype=np.uint8))]
    g = _gen(1)
    cd, nt = TINY_I2V["cross_attention_dim"], TINY_I2V["text_tokens"]
    return dict(frames=frames, images=images, prompt_embeds=torch.randn(1, nt, cd, generator=g), negative_prompt_embeds=torch.randn(1, nt, cd, generator=g))


# ---- the autoregressive outer loop (code/diffusion_trainer/streaming_svd.py:293-356) with a stand-in chunk generator ----
TINY_AR = dict(H=16, W=24, T=25, Tc=7, anchor=6, n_ar=2)


def tiny_ar_chunk0():
    """Chunk 0 as image_to_video hands it over (:388-394): PIL uint8 frames -> ToTensor -> * 2 - 1."""
    g = _gen(606)
    u8 = (torch.rand(TINY_AR["T"], 3, TINY_AR["H"], TINY_AR["W"], generator=g) * 255).round()
    return u8 / 255.0 * 2.0 - 1


def tiny_ar_generate(svd_input_frame, ctrl_frames, k):
    """Stand-in for _generate_conditional_output: T frames in [-1, 1] that depend on the anchor frame, on every control frame (with its
    position) and on the chunk counter -- any mistake in which frames are handed over changes the result."""
    T = TINY_AR["T"]
    ctrl = ctrl_frames[0]                                                   # [Tc, 3, H, W]
    w = torch.linspace(0.5, 1.5, ctrl.shape[0]).view(-1, 1, 1, 1)
    base = 0.6 * svd_input_frame + 0.4 * (ctrl * w).mean(0)
    ramp = torch.linspace(-0.3, 0.3, T).view(T, 1, 1, 1) * (1 + 0.1 * k)
    return torch.clamp(base[None] + ramp + 0.05 * torch.sin(7.0 * base[None] + k), -1.0, 1.0)


# ---- i2v_enhance_interface.vfi_process (:30-61) around a stand-in interpolator ----
def tiny_vfi_process_inputs():
    """6 uint8 RGB frames [36, 64, 3] hitting every byte value (the pass-through round trip k / 255. -> fp32 -> * 255 -> uint8 loses some k)."""
    import numpy as np
    rs = np.random.default_rng(21)
    frames = [rs.integers(0, 256, (36, 64, 3), dtype=np.uint8) for _ in range(6)]
    frames[0].reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    return frames


def tiny_vfi_process_infer(I0, I2):
    """stand-in for vfi.inference: asymmetric in (I0, I2) and in the channel order, values in [0, 1]; [1, 3, H, W] -> [1, 3, H, W]."""
    w = torch.tensor([0.2, 0.5, 0.9]).view(1, 3, 1, 1)
    return (0.35 * I0 + 0.65 * I2) * w + (1 - w) * I0.flip(3) * 0.5


# ---- i2v_enhance_interface.i2v_enhance_process (:86-138) around a recording stand-in pipeline ----
def tiny_enhance_process_inputs(n_frames):
    """([key image], n_frames distinct frames) as PIL images of 8 x 12 pixels (the function never looks at the pixels)."""
    import numpy as np
    import PIL.Image
    rs = np.random.default_rng(77)
    mk = lambda: PIL.Image.fromarray(rs.integers(0, 256, (8, 12, 3), dtype=np.uint8))
    return [mk()], [mk() for _ in range(n_frames)]


# ---- inference_i2v.StreamingPipeline.enhance_video (:192-209): image handling in front of the enhancer ----
def tiny_frontend_enhance_inputs():
    """(key image uint8 [576, 1024, 3], video uint8 [3, 576, 1024, 3]): smooth content so that the resampling filter matters."""
    g = _gen(909)
    low = torch.rand(4, 3, 18, 32, generator=g)
    up = torch.nn.functional.interpolate(low, size=(576, 1024), mode="bicubic", align_corners=False).clamp(0, 1)
    u8 = (up * 255).round().to(torch.uint8).permute(0, 2, 3, 1).numpy()
    return u8[0], u8[1:]


# ---- stage-1 conditioning (streaming_svd.py:155-221 + GeneralConditioner, config.yaml:160-218) with linear stand-ins for the networks ----
TINY_SVD_COND = dict(H=32, W=48, T=25, Tc=7, seed=4711)


def fake_clip_embed(img):
    """stand-in for the OpenCLIP image tower: [b, 3, H, W] in [-1, 1] -> [b, 1024]."""
    g = _gen(1001)
    proj = torch.randn(3 * 4 * 4, 1024, generator=g) * 0.2
    return torch.nn.functional.adaptive_avg_pool2d(img.float(), 4).flatten(1) @ proj.to(img.device)


def fake_cond_encode(x):
    """stand-in for AutoencoderKLModeOnly.encode: [b, 3, H, W] -> [b, 4, H/8, W/8]."""
    g = _gen(1002)
    mix = (torch.randn(4, 3, generator=g) * 0.7).to(x.device)
    return torch.einsum("oc,bchw->bohw", mix, torch.nn.functional.avg_pool2d(x.float(), 8))


def tiny_svd_cond_inputs():
    g = _gen(1003)
    c = TINY_SVD_COND
    return dict(frame=torch.rand(3, c["H"], c["W"], generator=g) * 2 - 1, ctrl_frames=torch.rand(1, c["Tc"], 3, c["H"], c["W"], generator=g) * 2 - 1)


# ---- the shipped architecture on a small latent (oracle/make_golden_fullarch.py, tools/fullarch_parity.py) ----
FULLARCH_CASE = dict(T=3, Tc=2, h=16, w=16, seed_unet=33, seed_cn=34)


def fullarch_inputs():
    g = _gen(5150)
    c = FULLARCH_CASE
    F, h, w = 2 * c["T"], c["h"], c["w"]
    return dict(x=torch.randn(F, 4, h, w, generator=g), t=torch.randn(F, generator=g) * 0.5, concat=torch.randn(F, 4, h, w, generator=g) * 0.5,
                crossattn=torch.randn(F, 1, 1024, generator=g), vector=torch.randn(F, 768, generator=g) * 0.5,
                ctrl_frames=torch.rand(1, c["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1)


# ---- the enhancer UNet at its shipped configuration on a small latent (oracle/make_golden_i2v_fullarch.py) ----
I2V_FULLARCH_CASE = dict(F=4, h=9, w=16, text_tokens=77, seed=6)


def i2v_fullarch_inputs():
    g = _gen(778)
    c = I2V_FULLARCH_CASE
    B, Fr, h, w, cd = 2, c["F"], c["h"], c["w"], 1024
    return dict(sample=torch.randn(B, 4, Fr, h, w, generator=g), t=torch.tensor(481), fps=torch.tensor([38, 38]),
                image_latents=torch.randn(B, 4, Fr, h, w, generator=g) * 0.7, image_embeddings=torch.randn(B, cd, generator=g),
                text=torch.randn(B, c["text_tokens"], cd, generator=g))


# ---- shipped-architecture cases of the smaller networks (oracle/make_golden_fullarch_small.py) ----
def fullarch_small_inputs():
    g = _gen(3131)
    low = torch.rand(1, 6, 64 // 8 + 2, 96 // 8 + 2, generator=g)
    pair = torch.nn.functional.interpolate(low, scale_factor=8, mode="bicubic", align_corners=False).clamp(0, 1)[:, :, 4:4 + 64, 7:7 + 96].contiguous()
    return dict(z=torch.randn(3, 4, 8, 8, generator=g), x_enc=torch.rand(1, 3, 64, 64, generator=g) * 2 - 1,
                img0=pair[:, :3].contiguous(), img1=pair[:, 3:].contiguous())


# ---- the shipped architecture at the SHIPPED PROBLEM SIZE (oracle/make_golden_fullsize.py; SURVEY 8a row A5 / A11 sizes) ----
FULLSIZE_CASE = dict(T=25, Tc=7, h=72, w=128, seed_unet=33, seed_cn=34)


def fullsize_inputs():
    """x is what Denoiser hands the network (x * c_in: unit variance); t = c_noise = 0.25 ln(sigma), one value per CFG half as in
    the sampler (sigma = 7.47 here, a mid-schedule AYS value); concat = cond-frame latents (0.18215-scaled VAE mean / zeros for uc)."""
    g = _gen(7272)
    c = FULLSIZE_CASE
    T, h, w = c["T"], c["h"], c["w"]
    F = 2 * T
    concat = torch.randn(1, 4, h, w, generator=g).mul(0.8).repeat(T, 1, 1, 1)
    return dict(x=torch.randn(F, 4, h, w, generator=g), t=torch.full((F,), 0.25 * 2.0112),
                concat=torch.cat((torch.zeros(T, 4, h, w), concat)),
                crossattn=torch.cat((torch.zeros(T, 1, 1024), torch.randn(1, 1, 1024, generator=g).repeat(T, 1, 1))),
                vector=(torch.randn(1, 768, generator=g) * 0.5).repeat(F, 1),
                ctrl_frames=torch.rand(1, c["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1)


# Round 5: the same forward at the two ends of the AYS-30 schedule, on fresh draws of every input (the round-4 golden sits at sigma = 7.47).
# c_noise, the timestep embedding and the activation statistics all move with sigma; x is what the Denoiser hands the network at that
# sigma: (latent + sigma * noise) * c_in with c_in = 1 / sqrt(sigma^2 + 1)  (denoiser_scaling.py:51-59).
FULLSIZE_SIGMA_CASES = {"s700": dict(sigma=700.0, seed=7575), "s0p063": dict(sigma=6.30212713e-02, seed=7676)}


def fullsize_inputs_sigma(name):
    import math
    cs = FULLSIZE_SIGMA_CASES[name]
    g = _gen(cs["seed"])
    c = FULLSIZE_CASE
    T, h, w = c["T"], c["h"], c["w"]
    F = 2 * T
    sigma = cs["sigma"]
    lat = torch.randn(T, 4, h, w, generator=g).mul(0.8)
    x = (lat + sigma * torch.randn(T, 4, h, w, generator=g)) / math.sqrt(sigma * sigma + 1.0)
    concat = torch.randn(1, 4, h, w, generator=g).mul(0.8).repeat(T, 1, 1, 1)
    return dict(x=torch.cat((x, x)), t=torch.full((F,), 0.25 * math.log(sigma)),
                concat=torch.cat((torch.zeros(T, 4, h, w), concat)),
                crossattn=torch.cat((torch.zeros(T, 1, 1024), torch.randn(1, 1, 1024, generator=g).repeat(T, 1, 1))),
                vector=(torch.randn(1, 768, generator=g) * 0.5).repeat(F, 1),
                ctrl_frames=torch.rand(1, c["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1)


# Round 5: one chunk's denoise + decode at the shipped size through the reference's own sampler / denoiser / guider / wrapper / decoder:
# 2 Euler steps of the AYS schedule (sigma 700 -> 0.002 -> 0: both ends of the schedule on one trajectory), guidance 1.5 -> 3.0, then
# decode_first_stage of the first group of 8 frames (oracle/make_golden_fullsize.py --which chunk).
FULLSIZE_CHUNK_CASE = dict(steps=2, decode_frames=8, seed=7777)


def fullsize_chunk_inputs():
    g = _gen(FULLSIZE_CHUNK_CASE["seed"])
    c = FULLSIZE_CASE
    T, h, w = c["T"], c["h"], c["w"]
    cond = dict(concat=torch.randn(1, 4, h, w, generator=g).mul(0.8).repeat(T, 1, 1, 1), crossattn=torch.randn(1, 1, 1024, generator=g).repeat(T, 1, 1),
                vector=(torch.randn(1, 768, generator=g) * 0.5).repeat(T, 1))
    uc = dict(concat=torch.zeros(T, 4, h, w), crossattn=torch.zeros(T, 1, 1024), vector=cond["vector"].clone())
    return dict(noise=torch.randn(T, 4, h, w, generator=g), c=cond, uc=uc, ctrl_frames=torch.rand(1, c["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1)


# Round 6: a WHOLE autoregressive chunk at the shipped size -- 30 AYS steps, all 25 frames decoded -- and the chunk after it, whose ctrl_frames are the
# last 7 DECODED frames of the first (streaming_svd.py:329-349).  The goldens come from oracle/make_golden_fullsize_gpu.py (the pinned restatement on the GPU
# box with stock PyTorch ops: the reference needs hours per chunk on CPU and cannot travel).  Fresh draws of every input; noise2 is the second chunk's noise.
FULLSIZE_CHUNK30_CASE = dict(steps=30, seed=7878)


def fullsize_chunk30_inputs():
    g = _gen(FULLSIZE_CHUNK30_CASE["seed"])
    c = FULLSIZE_CASE
    T, h, w = c["T"], c["h"], c["w"]
    cond = dict(concat=torch.randn(1, 4, h, w, generator=g).mul(0.8).repeat(T, 1, 1, 1), crossattn=torch.randn(1, 1, 1024, generator=g).repeat(T, 1, 1),
                vector=(torch.randn(1, 768, generator=g) * 0.5).repeat(T, 1))
    uc = dict(concat=torch.zeros(T, 4, h, w), crossattn=torch.zeros(T, 1, 1024), vector=cond["vector"].clone())
    return dict(noise=torch.randn(T, 4, h, w, generator=g), noise2=torch.randn(T, 4, h, w, generator=g), c=cond, uc=uc,
                ctrl_frames=torch.rand(1, c["Tc"], 3, 8 * h, 8 * w, generator=g) * 2 - 1)


# Round 5: the enhancer UNet at its shipped configuration AND latent size (90 x 160 = 720 x 1280 pixels; N = 14 400 spatial attention),
# CFG 2 x 4 frames (oracle/make_golden_i2v_fullarch.py --fullres)
I2V_FULLRES_CASE = dict(F=4, h=90, w=160, text_tokens=77, seed=6)


def i2v_fullres_inputs():
    g = _gen(779)
    c = I2V_FULLRES_CASE
    B, Fr, h, w, cd = 2, c["F"], c["h"], c["w"], 1024
    return dict(sample=torch.randn(B, 4, Fr, h, w, generator=g), t=torch.tensor(481), fps=torch.tensor([38, 38]),
                image_latents=torch.randn(B, 4, Fr, h, w, generator=g) * 0.7, image_embeddings=torch.randn(B, cd, generator=g),
                text=torch.randn(B, c["text_tokens"], cd, generator=g))


def fullsize_vae_inputs():
    g = _gen(7373)
    return dict(z=torch.randn(2, 4, 72, 128, generator=g))


def fullsize_pixel_subset(n_pixels):
    """seeded random 1/16 of the pixel positions of a frame (sorted), shared by generator and test."""
    return torch.randperm(n_pixels, generator=_gen(7474))[: n_pixels // 16].sort().values


# ---- appearance-preservation module (APM): SpatialVideoTransformer(use_apm=True) on a 17-token context (oracle/make_golden_apm.py) ----
def apm_inputs():
    g = _gen(1717)
    C, T, B, H, W = 320, 4, 2, 8, 8
    return dict(C=C, T=T, seed=17, x=torch.randn(B * T, C, H, W, generator=g),
                context=torch.randn(B, 1, 17, 1024, generator=g).repeat(1, T, 1, 1).reshape(B * T, 17, 1024) * 0.7)
