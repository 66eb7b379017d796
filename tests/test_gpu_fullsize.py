"""Full-size (BASELINE configs[1]: 2 x 25 frames @ latent 72x128, full 1.59 B-parameter VideoUNet) checks through size-independent
properties -- the CPU oracle needs ~7 minutes per forward at this size, so parity here is structural:

  * determinism: the path has no atomics; two runs are bit-identical;
  * CFG-batch independence: nothing in the forward couples the two CFG halves (per-sample GroupNorm statistics, per-frame and
    per-pixel attention) => forward(batch 2) == concat(forward(uncond half), forward(cond half)) bit for bit -- this is also what
    makes the CFG-pair split over two GPUs exact;
  * GEMM tile-configuration independence: every configuration accumulates K in the same order => identical bits at M = 460 800;
  * attention: softmax rows sum to 1 (V = 1 => O = 1) and the result does not depend on the order of the keys (up to
    reassociation noise) at N = 9216.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_unet():
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    ops.set_element_dtype(None)
    unet = VideoUNet(UNetConfig())
    unet.load_state_dict(init_by_name(unet.spec(), seed=33, device="cuda"), device="cuda")
    torch.cuda.empty_cache()
    return StreamingWrapper(unet, None, 7)


def _inputs(T=25, h=72, w=128):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")
    x, t = r(2 * T, 4, h, w), r(2 * T) * 0.5
    c = dict(concat=r(2 * T, 4, h, w) * 0.5, crossattn=r(2, 1, 1024).repeat_interleave(T, 0), vector=r(2, 768).repeat_interleave(T, 0) * 0.5)
    return x, t, c


def test_full_size_forward_deterministic_and_cfg_halves_independent(full_unet):
    T = 25
    x, t, c = _inputs(T)
    kw = dict(num_video_frames=T, ctrl_frames=None)
    with torch.no_grad():
        a = full_unet.forward(x, t, c, batch_size=2, image_only_indicator=torch.zeros(2, T, device="cuda"), **kw)
        b = full_unet.forward(x, t, c, batch_size=2, image_only_indicator=torch.zeros(2, T, device="cuda"), **kw)
        halves = [full_unet.forward(x[s], t[s], {k: v[s] for k, v in c.items()}, batch_size=1,
                                    image_only_indicator=torch.zeros(1, T, device="cuda"), **kw)
                  for s in (slice(0, T), slice(T, 2 * T))]
    assert a.shape == (2 * T, 4, 72, 128) and torch.isfinite(a).all() and a.float().std() > 1e-3
    assert torch.equal(a, b), "forward is not deterministic"
    assert torch.equal(a, torch.cat(halves, 0)), "the CFG halves are not independent"


def test_full_size_gemm_tile_config_independence():
    from streamingt2v_amd import ops
    from streamingt2v_amd.video_model import pack_geglu
    ops.set_element_dtype(None)
    M, K, N = 460800, 320, 2560
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    a = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5)
    b = torch.randn(N, generator=g, device="cuda")
    wp, bp = pack_geglu(w, b)
    wp = wp.to(torch.bfloat16)
    ref = ops.gemm(a, wp, bias=bp, geglu=True, tile_cfg=1)
    for cfg in (2, 8, 9, 17, 18, 19, 20):
        assert torch.equal(ops.gemm(a, wp, bias=bp, geglu=True, tile_cfg=cfg), ref), f"tile config {cfg} changes the bits"
    r = torch.randn(M, 320, generator=g, device="cuda").to(torch.bfloat16)
    w2 = (torch.randn(320, 1280, generator=g, device="cuda") * 1280 ** -0.5).to(torch.bfloat16)
    ref = ops.gemm(ref, w2, bias=b[:320].contiguous(), residual=r, tile_cfg=1)
    x = ops.gemm(a, wp, bias=bp, geglu=True, tile_cfg=8)
    for cfg in (2, 4, 8, 13, 17, 20):
        assert torch.equal(ops.gemm(x, w2, bias=b[:320].contiguous(), residual=r, tile_cfg=cfg), ref), f"tile config {cfg} changes the bits"


def test_full_size_attention_properties():
    from streamingt2v_amd import ops
    ops.set_element_dtype(None)
    frames, n, heads = 4, 9216, 5
    C = heads * 64
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    q = torch.randn(frames * n, C, generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn(frames * n, C, generator=g, device="cuda").to(torch.bfloat16)
    ones_t = torch.ones(frames, C, n, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(q)
    ops.attn_spatial(q, k, ones_t, out, frames, n, heads)
    assert (out.float() - 1).abs().max().item() <= 2 ** -7, "softmax rows do not sum to 1"
    v = torch.randn(frames, n, C, generator=g, device="cuda").to(torch.bfloat16)
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    ops.attn_spatial(q, k, v.transpose(1, 2).contiguous(), o1, frames, n, heads)
    perm = torch.randperm(n, generator=g, device="cuda")
    kp = k.view(frames, n, C)[:, perm].reshape(frames * n, C).contiguous()
    vp = v[:, perm].transpose(1, 2).contiguous()
    ops.attn_spatial(q, kp, vp, o2, frames, n, heads)
    assert (o1.float() - o2.float()).abs().max().item() <= 2e-3, "attention depends on the key order"


def test_full_size_enhancer_deterministic_and_cfg_halves_independent():
    """I2VGen-XL enhancer at the shipped size (2 x 38 frames @ latent 90x160, 1.42 B parameters): bit-identical reruns and
    batch-2 == two batch-1 forwards (the CFG halves are independent: per-sample GroupNorm, per-frame / per-pixel attention)."""
    from streamingt2v_amd import ops
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    ops.set_element_dtype(None)
    unet = I2VGenXLUNet(I2VConfig())
    unet.load_state_dict(init_by_name(unet.spec(), seed=5, device="cuda"), device="cuda")
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")
    Fr, H, W = 38, 90, 160
    sample, il, emb, text = r(2, 4, Fr, H, W), r(2, 4, Fr, H, W) * 0.7, r(2, 1024), r(2, 77, 1024)
    fps = torch.tensor([16, 16])
    with torch.no_grad():
        a = unet(sample, 481, fps=fps, image_latents=il, image_embeddings=emb, encoder_hidden_states=text)[0].clone()
        b = unet(sample, 481, fps=fps, image_latents=il, image_embeddings=emb, encoder_hidden_states=text)[0].clone()
        halves = [unet(sample[i:i + 1], 481, fps=fps[i:i + 1], image_latents=il[i:i + 1], image_embeddings=emb[i:i + 1],
                       encoder_hidden_states=text[i:i + 1])[0].clone() for i in range(2)]
    assert a.shape == (2, 4, Fr, H, W) and torch.isfinite(a).all() and a.float().std() > 1e-3
    assert torch.equal(a, b), "enhancer forward is not deterministic"
    assert torch.equal(a, torch.cat(halves, 0)), "the enhancer's CFG halves are not independent"


def test_full_size_vae_decode_deterministic():
    """Temporal-VAE decode of one 8-frame group at 576x1024 (1.2 GB activations at the top level): bit-identical reruns, finite,
    and sensitive to its input (every frame depends on every latent frame of the group: the time-stack GroupNorms pool their
    statistics over the group's frames)."""
    from streamingt2v_amd import ops
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VideoDecoder
    ops.set_element_dtype(None)
    dec = VideoDecoder()
    dec.load_state_dict(init_by_name(dec.spec(), seed=35, device="cuda"), device="cuda")
    vae = AutoencodingEngineDecoder(dec)
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    z = torch.randn(8, 4, 72, 128, generator=g, device="cuda")
    with torch.no_grad():
        a = vae.decode(z, timesteps=8).clone()
        b = vae.decode(z, timesteps=8).clone()
        z2 = z.clone(); z2[7] += 1.0
        c = vae.decode(z2, timesteps=8).clone()
    assert a.shape == (8, 3, 576, 1024) and torch.isfinite(a).all()
    assert torch.equal(a, b), "decode is not deterministic"
    assert not torch.equal(a[7], c[7]) and not torch.equal(a[0], c[0])


def test_full_size_vfi_rotation_equivariance_and_determinism():
    """EMA-VFI at the shipped size (F = 32, 720x1280, 65.7 M parameters), where the CPU oracle needs ~a minute per pair: fast TTA averages
    the prediction of the pair with the back-rotated prediction of the 180-degree rotated pair (Trainer.py:90-94), so interpolating the
    rotated frames must give EXACTLY the rotated middle frame (the two batch rows swap roles; fp32 addition commutes) -- this holds only if
    every kernel treats batch rows independently and deterministically.  Also: two runs are bit-identical, the uint8 frame is the
    truncation of the fp32 one, and the result stays inside [0, 1]."""
    from streamingt2v_amd import ops
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.params import init_by_name
    ops.set_element_dtype(None)
    torch.set_grad_enabled(False)
    m = EMAVFI(VFIConfig())
    m.load_state_dict(init_by_name(m.spec(), seed=3), device="cuda")
    g = torch.Generator().manual_seed(5)
    low = torch.rand(2, 3, 45, 80, generator=g)
    f0, f1 = (torch.nn.functional.interpolate(low[i:i + 1], size=(720, 1280), mode="bicubic").clamp(0, 1)[0].permute(1, 2, 0).contiguous().cuda()
              for i in range(2))
    a, a8 = m.inference(f0, f1, want_uint8=True)
    b, b8 = m.inference(f0, f1, want_uint8=True)
    assert torch.equal(a, b) and torch.equal(a8, b8), "EMA-VFI is not deterministic"
    rot = lambda t: t.flip(0).flip(1).contiguous()
    r, _ = m.inference(rot(f0), rot(f1), want_uint8=True)
    assert torch.equal(rot(r), a), f"rotation equivariance broken: max diff {(rot(r) - a).abs().max().item():.3e}"
    assert torch.isfinite(a).all() and a.min() >= 0 and a.max() <= 1
    assert torch.equal(a8, (a * 255.0).to(torch.uint8))
    print(f"[vfi full size] mean {a.mean().item():.4f}, |mid - (f0+f1)/2| mean {(a - (f0 + f1) / 2).abs().mean().item():.4f}")
