"""Parity AT THE SHIPPED ARCHITECTURE AND PROBLEM SIZE against the REFERENCE's own outputs (-m gpu).

Goldens: tests/golden/wrapper_fullsize.pt, vae_fullsize.pt (oracle/make_golden_fullsize.py: the unmodified reference modules on CPU, fp32:
StreamingWrapper.forward on CFG 2 x 25 frames @ 72x128 latent with ControlNet on 2 x 7 control frames of 576x1024; VideoDecoder on 2 frames
-> 576x1024), wrapper_fullarch.pt / i2v_fullarch.pt / vae_fullarch.pt / vae_enc_fullarch.pt (shipped architecture on a small latent).

Tolerance statement (absolute per-frame L2 = RMS error of a frame; values measured on MI355X, profiles/r04_fullsize_parity_plans.txt, _level0.txt):
  * row A11 (decoder, full size): fp16 8.9e-4 (round 6, decoder precision plan: 5.2e-4) -> asserted <= 1e-3, north_star's bound.
  * row A5 (StreamingWrapper.forward, full size, fp16), the PACKAGE DEFAULT (round 4: the precision plan of streamingt2v_amd/ops.py -- split-3 rim
    + fp32 residual stream in both networks): 0.757e-3 mean / 0.911e-3 max -> asserted mean <= 1e-3 AND max <= 1e-3: north_star's literal bound on
    every frame, in the configuration bench.py times.
    The cheaper plans, for the record (same file): everything 16 bit (rounds 2 / 3) 1.151e-3 / 1.377e-3; rim + ControlNet stream 1.018e-3 / 1.201e-3
    (+1.1 % forward time); + UNet stream at >= 640 channels 0.940e-3 / 1.116e-3 (+3.6 %); + level-0 ResBlocks 0.829e-3 / 1.002e-3.  The 16-bit
    configuration stays selectable (set_precision_plan(False, False, 0)) and is asserted inside the reference's own fp16-autocast envelope
    (tests/golden/wrapper_fullsize_autocast.json: 1.418e-3 / 1.675e-3, the unmodified reference under torch.autocast(float16), config.yaml:8).
  * bf16 (selectable, not the default): 8x coarser rounding, asserted <= 1.5e-2 / 1e-2.
Round 5 (profiles/r05_fullsize_parity_sigmas.txt, r05_i2v_parity_plans.txt; goldens from oracle/make_golden_fullsize.py --which round5 and
oracle/make_golden_i2v_fullarch.py --fullres, the unmodified reference modules on CPU):
  * row A5 at BOTH ENDS of the AYS schedule on fresh draws (sigma 700: c_noise 1.64, input scaled by 1/700; sigma 0.063: c_noise -0.69) next to the
    round-2 case (sigma 7.47): worst frame 0.935e-3 / 0.918e-3 / 0.917e-3, mean 0.76-0.78e-3 -> all three asserted <= 1e-3 on every frame.
  * row A12 (I2VGenXLUNet.forward, the enhancer) under ITS precision plan (ops.I2V_EXACT_RIM, I2V_STREAM_F32_MIN_CH = 320, default): shipped
    architecture on a 9x16 latent 0.907e-3, and at the SHIPPED latent size 90x160 (N = 14 400 spatial attention) 0.916e-3 max / 0.900e-3 mean ->
    asserted <= 1e-3 (round 4 without the plan: 1.40e-3, asserted <= 1.8e-3).
  * rows A1 + A3 + A4 + A5 + A11 composed: 2 Euler steps (sigma 700 -> 0.002 -> 0) of the reference's own sampler / denoiser / guider around the wrapper
    + decode of 8 frames at 576x1024: decoded frames 1.30e-3 max / 1.22e-3 mean, latents 2.7e-3 / 1.9e-3.  That is the network's per-evaluation
    deviation (0.76e-3 at sigma 700, where c_out = -1 and the denoised latent IS the network output) multiplied by classifier-free guidance:
    D = D_u + s (D_c - D_u) carries independent errors of both halves, sqrt(s^2 + (s - 1)^2) = 1.6 .. 3.6 for s = 1.5 .. 3.0 over the 25 frames
    (mean 2.5 -> 1.9e-3 on the latents, exactly what is measured), and the decoder's contraction.  No 16-bit-operand execution can be closer
    (the reference's own fp16 autocast deviates 1.42e-3 per evaluation, wrapper_fullsize_autocast.json).
Round 6: the composed chunk is asserted against the ENVELOPE of the reference's shipped precision instead of hand-set numbers -- tests/golden/chunk_fullsize_autocast.json
(oracle/make_golden_fullsize_gpu.py: fp16-autocast execution against fp32 of the same computation; 2-step chunk: frames 1.72e-3 max / 1.58e-3 mean, latents 4.86e-3 / 3.35e-3) --
and a WHOLE chunk (30 steps, 25 decoded frames) plus the autoregressive hand-over into the next chunk are compared the same way (test_chunk_full_size_vs_reference).
The decoder's precision plan (ops.AE_*) exists because of that comparison: with the 16-bit decoder the latents were 22 % inside the envelope and the decoded frames' mean 3 %
outside it (profiles/r06_fullsize_chunk_tests.txt) -- the reference decodes in fp32 (config.yaml:310).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
_SDS = {}


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_decoder_full_size_vs_reference(dtype):
    from tools.fullsize_parity import decoder_fullsize
    r = decoder_fullsize(dtype)
    print(f"[full-size VideoDecoder vs reference, {dtype}] per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} rel {r['rel_max']:.3e} corr {r['corr']:.7f}")
    assert r["abs_max"] <= (1e-3 if dtype == "fp16" else 1e-2), r
    assert r["corr"] >= (0.999995 if dtype == "fp16" else 0.9995)


@pytest.mark.parametrize("dtype,plan,case", [("fp16", "default", "sigma7.47"), ("fp16", "default", "s700"), ("fp16", "default", "s0p063"),
                                             ("fp16", "16bit", "sigma7.47"), ("bf16", "default", "sigma7.47")])
def test_streaming_wrapper_full_size_vs_reference(dtype, plan, case):
    from streamingt2v_amd import ops
    from tools.fullsize_parity import wrapper_fullsize
    import json
    import os
    # package defaults: fp16 elements, the round-4 precision plan (exact rim + fp32 residual stream in the ControlNet and in every UNet block)
    assert ops.DEFAULT_ELEM == torch.float16 and ops.EXACT_RIM and ops.CN_STREAM_F32 and ops.STREAM_F32_MIN_CH == 320 and not ops.STREAM_F32
    env = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrapper_fullsize_autocast.json")))["autocast_float16"]
    r = wrapper_fullsize(dtype, sds=_SDS, plan=None if plan == "default" else (False, False, 0), case=case)
    print(f"[full-size StreamingWrapper.forward vs reference, case {case}, {dtype}, precision plan {plan}] per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} "
          f"rel {r['rel_max']:.3e} corr {r['corr']:.7f}")
    if dtype == "fp16" and plan == "default":
        assert r["abs_mean"] <= 1e-3 and r["abs_max"] <= 1e-3, r          # north_star: per-frame L2 <= 1e-3, every frame, in the benchmarked configuration
        assert r["corr"] >= 0.999995
    elif dtype == "fp16":                                                 # all 16 bit (rounds 2 / 3): within the reference's own fp16-autocast envelope
        assert r["abs_mean"] <= env["l2_mean"] and r["abs_max"] <= env["l2_max"], (r, env)
        assert r["corr"] >= 0.999995
    else:
        assert r["abs_max"] <= 1.5e-2, r
        assert r["corr"] >= 0.9995


def test_chunk_full_size_vs_reference():
    """Rows A1 / A2 at the shipped size on DECODED FRAMES: sampler o denoiser o guider o StreamingWrapper o VideoDecoder
      * 2 Euler steps + decode of 8 frames against the REFERENCE's own CPU fp32 run (tests/golden/chunk_fullsize.pt);
      * a WHOLE chunk -- 30 AYS steps, all 25 frames decoded -- and the chunk after it, started from the last 7 frames THIS path decoded, against the pinned
        restatement run in fp32 on the MI355X with stock PyTorch ops (tests/golden/chunk30_fullsize.pt, ar_handover_fullsize.pt; the generator re-pins itself
        against the reference's committed CPU outputs to 3e-6 / 1e-5 before it writes: profiles/r06_golden_fullsize_gpu_log.txt).
    Bound: the ENVELOPE tests/golden/chunk_fullsize_autocast.json -- the same three computations with every network evaluation under torch.autocast(float16), i.e. the
    way the reference's shipped `precision: 16-mixed` (config.yaml:8) executes them, measured against their fp32 twins.  The HIP path must be at least as close to the
    fp32 result as that, on the worst frame and on average.  (Classifier-free guidance multiplies the per-evaluation deviation by sqrt(s^2 + (s - 1)^2) = 1.6 .. 3.6:
    no 16-bit-operand execution of a CHUNK holds the per-evaluation 1e-3; the literal numbers are printed.)"""
    import json
    import os
    from tools.fullsize_parity import chunk30_fullsize
    env = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chunk_fullsize_autocast.json")))
    r = chunk30_fullsize("fp16", sds=_SDS)
    _SDS.clear()          # 9 GB of host parameters: the last user of the shared weights
    for name, key in (("chunk2", "autocast_float16"), ("chunk30", "autocast_float16_chunk30"), ("handover", "autocast_float16_ar_handover")):
        f, z, e = r[name]["frames"], r[name]["z"], env[key]
        print(f"[full-size {name}: HIP fp16 vs fp32 reference] decoded frames per-frame L2 max {f['abs_max']:.3e} mean {f['abs_mean']:.3e} corr {f['corr']:.7f} | latents z max "
              f"{z['abs_max']:.3e} mean {z['abs_mean']:.3e}  ||  reference-precision envelope (fp16 autocast): frames max {e['frames_l2_max']:.3e} mean {e['frames_l2_mean']:.3e} | "
              f"z max {e['z_l2_max']:.3e} mean {e['z_l2_mean']:.3e}")
    print(f"[hand-over] the 7 control frames this path hands to the next chunk vs the fp32 run's: per-frame L2 max {r['handover_ctrl']['abs_max']:.3e}")
    for name, key in (("chunk2", "autocast_float16"), ("chunk30", "autocast_float16_chunk30"), ("handover", "autocast_float16_ar_handover")):
        f, z, e = r[name]["frames"], r[name]["z"], env[key]
        assert f["abs_max"] <= e["frames_l2_max"] and f["abs_mean"] <= e["frames_l2_mean"], (name, f, e)
        assert z["abs_max"] <= e["z_l2_max"] and z["abs_mean"] <= e["z_l2_mean"], (name, z, e)
        assert f["corr"] >= 0.99995, (name, f)


def test_enhancer_unet_full_resolution_vs_reference():
    """Row A12 at the SHIPPED latent size: I2VGenXLUNet.forward on CFG 2 x 4 frames @ 90x160 (N = 14 400 spatial attention tokens per frame) against
    the vendored module's fp32 output (tests/golden/i2v_fullres.pt), under the enhancer's default precision plan."""
    from streamingt2v_amd import ops
    from tools.i2v_parity import enhancer_parity
    assert ops.I2V_EXACT_RIM and ops.I2V_STREAM_F32_MIN_CH == 320
    r = enhancer_parity(None, fullres=True)
    print(f"[I2VGenXLUNet.forward 2x4 frames @ 90x160 vs vendored reference, fp16, default plan] per-frame L2 abs max {r['abs_max']:.3e} mean {r['abs_mean']:.3e} rel max {r['rel_max']:.3e}")
    assert r["abs_max"] <= 1e-3, r


def _err(out, ref):
    out, ref = out.float().cpu(), ref.float()
    return (out - ref).flatten(1).pow(2).mean(1).sqrt().max().item()


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_shipped_architecture_small_latent_vs_reference(dtype, golden_dir):
    """The 4-level StreamingWrapper (1.59 B + 0.67 B parameters), the enhancer's I2VGenXLUNet (1.42 B), the temporal VideoDecoder and the
    sgm Encoder at their shipped widths on small latents, each against the unmodified reference module's output."""
    import os
    from oracle.cases import FULLARCH_CASE as c, I2V_FULLARCH_CASE as ci, fullarch_inputs, fullarch_small_inputs, i2v_fullarch_inputs
    from streamingt2v_amd import ops
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import Encoder, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    torch.set_grad_enabled(False)
    f16 = dtype == "fp16"
    ops.set_element_dtype(torch.float16 if f16 else torch.bfloat16)
    try:
        si = fullarch_small_inputs()
        dec = VideoDecoder(); dec.load_state_dict(init_by_name(dec.spec(), seed=35), device="cuda")
        e_dec = _err(dec.forward(si["z"].cuda(), timesteps=3), torch.load(os.path.join(golden_dir, "vae_fullarch.pt"))["out"])
        enc = Encoder(); enc.load_state_dict(init_by_name(enc.spec(), seed=36), device="cuda")
        e_enc = _err(enc(si["x_enc"].cuda()), torch.load(os.path.join(golden_dir, "vae_enc_fullarch.pt"))["out"])
        del dec, enc
        eu = I2VGenXLUNet(I2VConfig()); eu.load_state_dict(init_by_name(eu.spec(), seed=ci["seed"]), device="cuda")
        ei = i2v_fullarch_inputs()
        fr = lambda x: x.permute(0, 2, 1, 3, 4).reshape(-1, *x.shape[1:2], *x.shape[3:])
        out = eu(ei["sample"], ei["t"], fps=ei["fps"], image_latents=ei["image_latents"], image_embeddings=ei["image_embeddings"], encoder_hidden_states=ei["text"])[0]
        e_i2v = _err(fr(out), fr(torch.load(os.path.join(golden_dir, "i2v_fullarch.pt"))["out"]))
        del eu
        torch.cuda.empty_cache()
        cfg = UNetConfig()
        unet, cn = VideoUNet(cfg), ControlNet(cfg)
        unet.load_state_dict(init_by_name(unet.spec(), seed=c["seed_unet"]), device="cuda")
        cn.load_state_dict(init_by_name(cn.spec(), seed=c["seed_cn"]), device="cuda")
        inp = {k: v.cuda() for k, v in fullarch_inputs().items()}
        T = c["T"]
        gold = torch.load(os.path.join(golden_dir, "wrapper_fullarch.pt"))
        out = StreamingWrapper(unet, cn, c["Tc"]).forward(inp["x"], inp["t"], {k: inp[k] for k in ("concat", "crossattn", "vector")}, batch_size=2,
                                                        num_video_frames=T, image_only_indicator=torch.zeros(2, T, device="cuda"), ctrl_frames=inp["ctrl_frames"])
        e_w = _err(out, gold["out"])
        out = unet.forward(torch.cat((inp["x"], inp["concat"]), 1), inp["t"], context=inp["crossattn"], y=inp["vector"], num_video_frames=T,
                           image_only_indicator=torch.zeros(2, T, device="cuda"))
        e_u = _err(out, gold["out_noctrl"])
        print(f"[shipped architecture, small latent, {dtype}] per-frame L2 abs max: StreamingWrapper {e_w:.3e} | VideoUNet (no control) {e_u:.3e} | "
              f"I2VGenXLUNet {e_i2v:.3e} | VideoDecoder {e_dec:.3e} | Encoder {e_enc:.3e}")
        k = 1.0 if f16 else 9.0          # bf16: one rounding is 8x coarser
        # StreamingWrapper (A5), VideoUNet without control (A7) and -- round 5 -- I2VGenXLUNet (A12) under their default precision plans: north_star's 1e-3
        # (measured 0.76e-3 / 0.69e-3 / 0.907e-3)
        # round 6: the VideoDecoder under its precision plan (ops.AE_EXACT_RIM, AE_STREAM_F32_MIN_CH = 128: 0.64e-3) and the Encoder (0.91e-3) hold the same bound
        assert e_w <= 1e-3 * k and e_u <= 1e-3 * k and e_i2v <= 1e-3 * k and e_dec <= 1e-3 * k and e_enc <= 1e-3 * k
    finally:
        ops.set_element_dtype(None)
        torch.cuda.empty_cache()
