#!/usr/bin/env python
"""bench.py -- generated frames/sec of the StreamingSVD hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload stage1|c2|ar_chunk|c3|enhance|vfi|full] [--dtype fp16|bf16]
    (--gpus N > 1 without a launcher re-executes itself as N ranks under torch.distributed.run; under a launcher it reads RANK / WORLD_SIZE)

DEFAULT workload = stage 1 of the 200-frame job (BASELINE.json configs[2]; SURVEY.md 8d "stage-1 frames/s"): 100 diffusion-stage
frames at 576x1024 = chunk 0 (25 frames, 25 Euler-EDM steps, no ControlNet) + 5 autoregressive chunks (30 AlignYourSteps steps,
ControlNet on the previous chunk's last 7 DECODED frames + 13 CAM mergers, 18 new frames each), every chunk decoded by the temporal
VAE; CFG batch 2 x 25 frames per network evaluation at latent 72x128.  One "step" = ONE CHUNK of that sequence, in order
(chunk 0, AR1 ... AR5, chunk 0 of the next video, ...): warm-up and timed steps walk the real sequence, so the control frames of
every AR chunk are the decoded frames of the chunk before it.  value = frames that end up in the 100-frame videos (25 | 18 x 4 | 3 per
chunk: the last AR chunk is cut by [:100], inference_i2v.py:190) / max-over-ranks time.
  c2        BASELINE.json configs[1]: SVD-XT UNet single chunk, 25 frames, 25 steps, no ControlNet/CAM (parity-test case).
  ar_chunk  one autoregressive chunk repeated (30 AYS steps, ControlNet + CAM, decode 25 frames).
  c3        one step = a whole 100-frame stage-1 video.       enhance / vfi: the I2VGen-XL window pass / EMA-VFI.
  full      BASELINE.json configs[4] / SURVEY 8d metric (ii): one step = one whole 200-frame job, stage 1 + enhancement with randomized
            blending + frame interpolation; value = 200 final frames / end-to-end seconds, per-stage seconds in config.
Weights: seeded random at the reference's exact architecture (no checkpoints offline; zero-inits un-zeroed).
Inputs already resident in HBM when the timed region starts.
Multi-GPU (--gpus N > 1): default --parallelism auto = job for the job-level workloads: ONE job strong-scaled over all GPUs (CFG pair -- the two CFG
halves of every network call on two ranks, one RCCL all-gather of the 3.7 MB network output per Euler step -- x frame<->pixel sequence parallelism of
degree N/2, RCCL all-to-all around the temporal operators; SURVEY 8e options 1 + 2): chunks of one video are sequential (chunk k+1 needs chunk k's
decoded frames), so a job scales INSIDE a chunk.  --parallelism pairs: N/2 independent videos, each on a CFG pair (throughput; weak scaling beyond 2 GPUs).
--parallelism replica: one independent video per GPU (no collective).

The JSON line also carries
  roofline     : dominant kernel (by summed device time) of a SEPARATE traced AR chunk (HIP events around every GEMM / attention
                 launch on the launch stream; the timed region itself is untraced): algorithmic FLOP / event time vs the 2.5 PFLOP/s
                 16-bit MFMA peak, per-signature HBM bytes (rocprofv3 PMC passes, profiles/) next to the algorithmic bytes.
  cpu_baseline : the CPU oracle (oracle/svd_oracle.py, a port of the reference path) timed on this host at the REAL spatial size
                 (72x128 latent, CFG 2 x 2 frames; one 576x1024 frame decoded), extrapolated linearly over frames only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAT_H, LAT_W, T_FRAMES = 72, 128, 25
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


class LaunchTrace:
    """Per-launch HIP-event timing on the stream the kernels run on (torch's current stream).  Installed as ops.trace for ONE
    separate chunk after the timed region (the timed region is untraced)."""

    def __init__(self):
        self.records = []      # (name, signature, flops, bytes, start_event, stop_event)

    class _Ctx:
        def __init__(self, tr, name, flops, sig, nbytes):
            self.tr, self.name, self.flops, self.sig, self.nbytes = tr, name, flops, sig, nbytes

        def __enter__(self):
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

        def __exit__(self, *a):
            self.e.record()
            self.tr.records.append((self.name, self.sig, self.flops, self.nbytes, self.s, self.e))

    def launch(self, name, flops, sig=None, nbytes=0.0):
        return LaunchTrace._Ctx(self, name, flops, sig or name, nbytes)

    def summarize(self):
        """{kernel name: [launches, flops, ms, {signature: [launches, flops, ms, algorithmic bytes per launch]}]}"""
        agg = {}
        for name, sig, flops, nbytes, s, e in self.records:
            ms = s.elapsed_time(e)
            a = agg.setdefault(name, [0, 0.0, 0.0, {}])
            a[0] += 1; a[1] += flops; a[2] += ms
            b = a[3].setdefault(sig, [0, 0.0, 0.0, nbytes])
            b[0] += 1; b[1] += flops; b[2] += ms
        return agg


def signature_traffic(sig):
    """HBM bytes per launch of ONE GEMM signature from the committed rocprofv3 PMC passes (tools/pmc_signature.sh: the signature is
    launched alone under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; gfx950 x2 read correction of
    MI355X_MICROARCH.md) -> profiles/*traffic_signatures*.json.  PMC cannot be sampled inside this process; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_signatures*.json")))
    for f in reversed(files):
        ent = json.load(open(f)).get("signatures", {}).get(sig)
        if ent:
            return dict(ent, source=os.path.basename(f))
    return None


def roofline_from_trace(trace):
    """The dominant kernel of the traced chunk.  All tile instantiations of the GEMM template (gemm_impl.inc) are ONE kernel family here:
    achieved = their summed algorithmic flops / their summed HIP-event time (per-instantiation lines stay in `traced_kernels`)."""
    agg = trace.summarize()
    tot_ms = sum(v[2] for v in agg.values())
    fam = {}
    for k, v in agg.items():
        f = "gemm_kernel<tile, view, elem> (all instantiations)" if k.startswith("gemm_cfg") else k
        a = fam.setdefault(f, [0, 0.0, 0.0, {}, 0])
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[4] += 1
        for sg, sv in v[3].items():
            b = a[3].setdefault(sg, [0, 0.0, 0.0, sv[3]])
            b[0] += sv[0]; b[1] += sv[1]; b[2] += sv[2]
    name, (cnt, flops, ms, sigs, ninst) = max(fam.items(), key=lambda kv: kv[1][2])
    ach = flops / (ms * 1e-3) / 1e12
    sig, (scnt, sflops, sms, sbytes) = max(sigs.items(), key=lambda kv: kv[1][2])
    counted = signature_traffic(sig)
    traffic = {"signature": sig, "launches": scnt, "avg_launch_ms": round(sms / scnt, 4), "TFLOP/s": round(sflops / (sms * 1e-3) / 1e12, 1),
               "algorithmic_bytes_per_launch": round(sbytes) if sbytes else None, "counted_bytes_per_launch": None, "counted_over_algorithmic": None}
    if counted and sbytes:
        traffic.update(counted_bytes_per_launch=round(counted["bytes_per_launch"]),
                       counted_over_algorithmic=round(counted["bytes_per_launch"] / max(sbytes, 1.0), 3), source=counted["source"],
                       note="FETCH_SIZE*2 (gfx950 correction) + WRITE_SIZE of this signature launched alone; algorithmic = A + W read once, C (+ residual) once")
    return {"bound": "mfma", "kernel": name, "instantiations": ninst, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "launches": cnt, "avg_launch_ms": round(ms / cnt, 4),
            "share_of_traced_time": round(ms / tot_ms, 3), "traced_ms_total": round(tot_ms, 1),
            "traced": "one step after the timed region; HIP events around every GEMM / MFMA-attention launch (norms, element-wise and temporal attention are not traced: share_of_traced_time is among the traced launches only)",
            "traced_kernels": {k: {"launches": v[0], "ms": round(v[2], 2), "TFLOP/s": round(v[1] / (v[2] * 1e-3) / 1e12, 1)}
                               for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}}


def precision_plan():
    """The round-4 precision plan the networks were loaded under (streamingt2v_amd.ops) -- part of every line's config."""
    from streamingt2v_amd import ops
    return {"element": str(ops.ELEM).replace("torch.", ""), "exact_rim": bool(ops.EXACT_RIM), "controlnet_stream_fp32": bool(ops.CN_STREAM_F32),
            "unet_stream_fp32_min_channels": int(ops.STREAM_F32_MIN_CH), "unet_stream_fp32_everywhere": bool(ops.STREAM_F32),
            "decoder_exact_rim": bool(ops.AE_EXACT_RIM), "decoder_stream_fp32_min_channels": int(ops.AE_STREAM_F32_MIN_CH)}


def kernel_switches():
    """The A/B switches of the kernel path (environment overrides in streamingt2v_amd.ops): an A/B line says which side it is."""
    from streamingt2v_amd import ops
    return {"ff_fused": bool(ops.FF_FUSED), "ff_fused_layernorm": bool(ops.FF_FUSED_LN), "rowgemm320": bool(ops.ROWGEMM),
            "zero_copy_concat": bool(ops.ZERO_COPY_CONCAT), "zero_copy_generic": bool(ops.ZERO_COPY_GENERIC)}


def build_models(workload, device):
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencodingEngineDecoder, VideoDecoder
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    use_cn = workload in ("ar_chunk", "stage1", "c3")
    cfg = UNetConfig(controlnet_mode=use_cn)
    unet = VideoUNet(cfg)
    unet.load_state_dict(init_by_name(unet.spec(), seed=33, device=device), device=device)
    cn = None
    if use_cn:
        cn = ControlNet(cfg)
        cn.load_state_dict(init_by_name(cn.spec(), seed=34, device=device), device=device)
    dec = VideoDecoder()
    dec.load_state_dict(init_by_name(dec.spec(), seed=35, device=device), device=device)
    torch.cuda.empty_cache()
    return StreamingWrapper(unet, cn, 7), AutoencodingEngineDecoder(dec)


def synthetic_inputs(device, seed):
    """seeded latents / conditioning of the shipped shapes."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    T, h, w = T_FRAMES, LAT_H, LAT_W
    r = lambda *s: torch.randn(*s, generator=g, device=device)
    c = dict(concat=r(1, 4, h, w).repeat(T, 1, 1, 1) * 0.18215 * 5, crossattn=r(1, 1, 1024).repeat(T, 1, 1),
             vector=r(1, 768).repeat(T, 1) * 0.5)
    uc = dict(concat=torch.zeros_like(c["concat"]), crossattn=torch.zeros_like(c["crossattn"]), vector=c["vector"].clone())
    ctrl = torch.rand(1, 7, 3, 8 * h, 8 * w, generator=g, device=device) * 2 - 1
    return c, uc, ctrl, r(T, 4, h, w)


def cpu_baseline(workload):
    """Time the CPU oracle (a port of the reference path, pinned against the unmodified reference: oracle/make_golden*.py) on a bounded
    sample at the REAL spatial size and extrapolate over FRAMES only (every operator is linear in the number of frames except the
    temporal attention / (3,1,1) convolutions, < 1 % of the FLOPs):
      * one full-architecture VideoUNet forward (1.59 B parameters, fp32) on CFG 2 x 2 frames at the 72x128 latent -- the N = 9216
        spatial attention, the 295-MB-class activations and the cache behaviour of the real forward are all in the sample;
      * one frame decoded by the temporal VAE at 576x1024.
    A forward of the job has 50 frames (x 12.5); AR chunks add ControlNet + CAM: x 181.96 / 159.9 algorithmic FLOP.
    kind "port": the reference is Python and cannot travel to the GPU box in any form, so what is timed HERE is its pinned restatement (same library
    calls: F.conv2d / F.linear / SDPA / group_norm in fp32).  The reference's own modules were timed once in the 8-core build container
    (553 s per forward: profiles/r02_cpu_reference_forward.txt); that number is documentation, not part of this line."""
    from oracle import svd_oracle as O
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import VideoDecoder
    from streamingt2v_amd.video_model import UNetConfig, VideoUNet
    cores = min(os.cpu_count() or 1, 32)      # more threads than this slow the fp32 CPU kernels down on a 256-core host
    torch.set_num_threads(cores)
    with torch.no_grad():
        sd = init_by_name(VideoUNet(UNetConfig(controlnet_mode=False)).spec(), seed=33)
        T, h, w = 2, LAT_H, LAT_W
        g = torch.Generator(); g.manual_seed(0)
        x = torch.randn(2 * T, 8, h, w, generator=g)
        t0 = time.time()
        O.video_unet(sd, O.Cfg(), x, torch.zeros(2 * T), torch.randn(2 * T, 1, 1024, generator=g),
                     torch.randn(2 * T, 768, generator=g), T)
        t_unet = time.time() - t0
        del sd
        sd = init_by_name(VideoDecoder().spec(), seed=35)
        t0 = time.time()
        O.video_decoder(sd, O.VaeCfg(), torch.randn(1, 4, h, w, generator=g), 1)
        t_dec = time.time() - t0
    fwd_c2 = t_unet * (2 * T_FRAMES) / (2 * T)                    # 50 frames
    fwd_ar = fwd_c2 * 181.96 / 159.9
    dec_chunk = T_FRAMES * t_dec
    if workload == "full":
        # + the enhancement stage: one I2VGen-XL UNet forward of the CPU oracle (full architecture, 1.42 B parameters) on CFG 2 x 2 frames at the
        # 90x160 latent, extrapolated over frames (x 19 for a 38-frame window), 29 DDIM steps, 3 blending windows + the 3-frame key-frame pre-pass
        # counted as 3/38 of a window; VAE encode / decode and EMA-VFI are not counted (they would make the baseline slower still).
        from oracle import i2vgen_oracle as IO
        from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
        with torch.no_grad():
            sd = init_by_name(I2VGenXLUNet(I2VConfig()).spec(), seed=5)
            Fe, he, we = 2, 90, 160
            xs = torch.randn(2, 4, Fe, he, we, generator=g)
            t0 = time.time()
            IO.unet(sd, xs, torch.tensor(500), torch.tensor([38, 38]), torch.randn(2, 4, Fe, he, we, generator=g), torch.randn(2, 1024, generator=g),
                    torch.randn(2, 77, 1024, generator=g))
            t_enh = time.time() - t0
            del sd
        win_fwd = t_enh * 38 / Fe
        enh_s = 29 * win_fwd * (3 + 3 / 38)
        stage1_s = (25 * fwd_c2 + dec_chunk) + 5 * (30 * fwd_ar + dec_chunk)
        return {"value": 180 / (stage1_s + enh_s), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"stage 1: oracle VideoUNet forward, CFG 2 x {T} frames @ {h}x{w} latent: {t_unet:.1f} s; VideoDecoder 1 frame @ 576x1024: {t_dec:.1f} s "
                          f"=> {stage1_s:.0f} s per 100-frame stage 1; enhancement: oracle I2VGenXLUNet forward, CFG 2 x {Fe} frames @ {he}x{we} latent: {t_enh:.1f} s; "
                          f"x19 over frames => {win_fwd:.0f} s per 38-frame window forward, x 29 DDIM steps x (3 windows + key-frame pre-pass) => {enh_s:.0f} s; "
                          f"VAE encode / decode and EMA-VFI not counted; 180 final frames per job"}
    if workload == "c2":
        chunk_s, frames, what = 25 * fwd_c2 + dec_chunk, T_FRAMES, "25-step chunk without ControlNet"
    elif workload == "ar_chunk":
        chunk_s, frames, what = 30 * fwd_ar + dec_chunk, T_FRAMES, "30-step AR chunk"
    else:
        chunk_s, frames, what = (25 * fwd_c2 + dec_chunk) + 5 * (30 * fwd_ar + dec_chunk), 100, "100-frame stage 1 (chunk 0 + 5 AR chunks)"
    return {"value": frames / chunk_s, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle VideoUNet forward, CFG 2 x {T} frames @ {h}x{w} latent (full size): {t_unet:.1f} s; VideoDecoder 1 frame @ 576x1024: "
                      f"{t_dec:.1f} s; x12.5 over frames => {fwd_c2:.0f} s per 50-frame forward (x1.138 with ControlNet + CAM), "
                      f"{dec_chunk:.0f} s per 25-frame decode => {chunk_s:.0f} s per {what}"}


def run_enhance(args, rank, world, device):
    """--workload enhance: one step = the whole SDEdit pass (29 DDIM steps, CFG 9) of ONE 38-frame window
    of the I2VGen-XL enhancer at latent 90x160 (720x1280 pixels); every rank takes its own window (the windows of a step are
    independent: weak scaling, the all-gather of blend_step_sharded moves 4.4 MB per window and is not part of this loop)."""
    from streamingt2v_amd import ops, parallel
    from streamingt2v_amd.enhance import DDIMSchedule, I2VEnhancer
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    chunk, H, W, cd = 38, 90, 160, 1024
    # N = 1: one window, no overlap (the per-window number of rounds 2 / 3).  N > 1: the SHIPPED job -- one 100-frame video = 3 blending windows of
    # 38 frames, overlap 12 (i2v_enhance_interface.py:92-118), 90 frames kept -- strong-scaled: every DDIM step its 3 x 2 (window, CFG half)
    # units go round-robin over the ranks and ONE all-gather of the units' predictions (8.75 MB each) closes the step
    # (blending.blend_step_units_sharded); 6 units keep 6 of 8 GPUs busy (whole windows: 3 of 8).
    n_win = 3 if world > 1 else 1
    overlap = 12 if world > 1 else 0
    n_frames = n_win * chunk - (n_win - 1) * overlap
    unet = I2VGenXLUNet(I2VConfig())
    unet.load_state_dict(init_by_name(unet.spec(), seed=5, device=device), device=device)
    g = torch.Generator(device=device); g.manual_seed(33)            # identical on every rank: all ranks hold the same video
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    conds = []
    for _ in range(n_win):
        il, emb, text = rn(1, 4, chunk, H, W) * 0.7, rn(1, cd), rn(1, 77, cd)
        conds.append(dict(fps=torch.tensor([38, 38]), image_latents=torch.cat([il, il]),
                          image_embeddings=torch.cat([torch.zeros_like(emb), emb]), text=torch.cat([torch.zeros_like(text), text])))
    video, noise = rn(1, 4, n_frames, H, W) * 0.5, rn(1, 4, n_frames, H, W)
    n_inf = args.denoise_steps or 30
    enh = I2VEnhancer(unet, DDIMSchedule(), guidance_scale=9.0, num_inference_steps=n_inf, strength=0.97)
    n_steps = len(DDIMSchedule().get_timesteps(n_inf, 0.97))
    group = eplan = None
    if world > 1:
        import torch.distributed as dist
        group = dist.group.WORLD
        if args.parallelism == "job" and world in (4, 8, 16):
            # round 5: the stage-1 job plan on the enhancer -- every window on ALL ranks (CFG pair x frame<->pixel sequence parallelism of degree world / 2 inside
            # I2VGenXLUNet; the 90x160 .. 12x20 levels and the 38 frames split over 2 / 4 / 8 ranks); the default stays the (window, CFG half) units
            eplan = parallel.JobPlan.from_env(world, "job", frames_cond=chunk, min_pix=240)
            if eplan.mode != "job" or eplan.sp is None:
                eplan = None

    def one():
        import random
        with torch.no_grad():
            kw = dict(plan=eplan) if eplan is not None else dict(group=group)
            return enh.denoise(video, noise, conds, chunk, overlap, rng=random.Random(33), **kw)

    def sync_all():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    trace = None if args.no_trace else LaunchTrace()
    ops.trace = trace
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    ops.trace = None
    assert torch.isfinite(out).all()
    if rank == 0:
        roof = roofline_from_trace(trace) if trace is not None else None
        print(json.dumps({
            "metric": "enhanced frames/sec (720x1280)", "value": round(args.steps * n_frames / dt, 4), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": ("I2VGen-XL enhancement (SDEdit): %d blending window(s) of 38 frames @ latent 90x160, overlap %d, CFG 9 (batch 2x38 per window), "
                                    "%d DDIM steps" % (n_win, overlap, n_steps)),
                       "frames_per_step": n_frames, "denoise_steps": n_steps, "latent": [H, W],
                       "parallelism": ((f"every window on all {world} GPUs: CFG pair x frame<->pixel sequence parallelism of degree {world // 2} inside I2VGenXLUNet" if eplan is not None else
                                        f"{2 * n_win} (window, CFG half) units round-robin over {world} GPUs, one all-gather of the units' predictions per DDIM step")
                                       if world > 1 else "single window"),
                       "weights": "seeded random, reference architecture (1.42 B parameters)"},
            "roofline": roof, "cpu_baseline": None}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_c3(args, rank, world, device):
    """--workload c3 (SURVEY.md 8d "stage-1 frames/s"): one step = a whole autoregressive video: chunk 0 (25 frames, 25 EDM steps,
    no ControlNet) + 5 AR chunks (30 AYS steps, ControlNet on the previous chunk's last 7 decoded frames + CAM, 18 new frames each)
    + temporal-VAE decode of every chunk -> 115 frames, cut to 100 (inference_i2v.py:190).  The per-chunk conditioner (CLIP image
    tower + VAE encoder) is not part of the path: synthetic conditioning tensors are reused for every chunk."""
    from streamingt2v_amd import ops, parallel
    from streamingt2v_amd.sampling import AlignYourSteps, EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    wrapper, vae = build_models("ar_chunk", device)
    sampler = EulerEDMSampler(num_steps=args.denoise_steps or 30, num_frames=T_FRAMES, min_scale=1.5, max_scale=3.0,
                              discretization=AlignYourSteps())
    model = StreamingSVD(wrapper, vae, sampler)
    c, uc, _, _ = synthetic_inputs(device, 33 + rank)
    g = torch.Generator(device=device); g.manual_seed(133 + rank)
    noises = [torch.randn(T_FRAMES, 4, LAT_H, LAT_W, generator=g, device=device) for _ in range(6)]
    image = torch.rand(3, 8 * LAT_H, 8 * LAT_W, generator=g, device=device) * 2 - 1
    n_frames = 100

    def one():
        with torch.no_grad():
            video = model.image_to_video(lambda frame: (c, uc), image, n_frames, noises, num_steps=args.denoise_steps)
            return model.to_uint8_video(video)

    def sync_all():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    assert out.shape[0] == n_frames and out.dtype == torch.uint8
    if rank == 0:
        print(json.dumps({
            "metric": "generated frames/sec (576x1024)", "value": round(world * args.steps * n_frames / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": "StreamingSVD stage 1, 100 frames @576x1024: chunk 0 (25 EDM steps) + 5 AR chunks (30 AYS steps, "
                                   "ControlNet(7) + CAM), CFG 2, temporal-VAE decode, uint8 frames", "frames_per_step": n_frames,
                       "latent": [LAT_H, LAT_W], "parallelism": f"replica-per-gpu x{world}", "weights": "seeded random, reference architecture"},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


class Stage1Stream:
    """Walks the chunk sequence of consecutive 100-frame stage-1 jobs: chunk 0, AR1 .. AR5, chunk 0 of the next video, ...
    (StreamingSVD.image_to_video + _autoregressive_generation, diffusion_trainer/streaming_svd.py:293-402, one chunk per step())."""
    KEPT = (25, 18, 18, 18, 18, 3)       # frames of each chunk that end up in video[:100] (inference_i2v.py:190)

    def __init__(self, model, c, uc, noises, start=0, prev_frames=None):
        """start: position in the 6-chunk cycle the first step() runs (bench: chosen so that the WARM-UP ends at a video boundary and the timed
        region starts with chunk 0); a start inside a video needs `prev_frames` [>=7, 3, H, W], the stand-in for the decoded previous chunk
        (used by warm-up steps only)."""
        self.model, self.c, self.uc, self.noises = model, c, uc, noises
        self.i, self.chunks, self.video_u8 = start % 6, None, None
        if self.i:
            assert prev_frames is not None
            self.chunks = [prev_frames]

    def step(self):
        m, k = self.model, self.i % 6
        with torch.no_grad():
            if k == 0:
                first = m.quantize_like_pil(m._generate_initial_chunk(self.c, self.uc, self.noises[0]))
                self.chunks = [first]
            else:
                ctrl = m.extract_ctrl_frames(self.chunks[-1], m.num_conditional_frames)
                result = m._generate_conditional_output(self.c, self.uc, ctrl, self.noises[k])
                self.chunks.append(result[m.num_conditional_frames:])
                if k == 5:
                    self.video_u8 = m.to_uint8_video(torch.cat(self.chunks, 0)[:100])
        self.i += 1
        return self.KEPT[k]


def build_stage1(args, world, device):
    """Models + chunk stream of the stage-1 workload under the job plan of this launch."""
    from streamingt2v_amd import parallel
    from streamingt2v_amd.sampling import AlignYourSteps, EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    plan = parallel.JobPlan.from_env(world, args.parallelism)          # CFG pair x sequence-parallel group (or replicas)
    wrapper, vae = build_models("stage1", device)
    plan.attach(wrapper, vae)
    sampler = EulerEDMSampler(num_steps=args.denoise_steps or 30, num_frames=T_FRAMES, min_scale=1.5, max_scale=3.0,
                              discretization=AlignYourSteps(), cfg_exchange=plan.cfg_exchange, use_graph=args.graph)
    model = StreamingSVD(wrapper, vae, sampler)
    model.use_graph = args.graph
    if args.denoise_steps:
        model.initial_num_steps = args.denoise_steps
    c, uc, _, _ = synthetic_inputs(device, 33 + plan.video_id)
    g = torch.Generator(device=device); g.manual_seed(133 + plan.video_id)
    noises = [torch.randn(T_FRAMES, 4, LAT_H, LAT_W, generator=g, device=device) for _ in range(6)]
    return plan, model, c, uc, noises, g


def other_stages(args, device, stage1_s):
    """The rest of the 200-frame job next to the driver-timed stage-1 number (single GPU, after the timed region, never inside it): ONE 38-frame window of the
    I2VGen-XL enhancer (29 DDIM steps, CFG 9, latent 90x160; pipeline_i2vgen_xl.py:841-913) with its own launch trace, and EMA-VFI frame pairs at 720x1280.
    The shipped job enhances 100 frames as 3 windows of 38 (overlap 12) + a 3-frame key-frame pre-pass and interpolates 99 pairs (i2v_enhance_interface.py:86-138,
    inference_i2v.py:211-224): final_frames_per_s_estimate = 180 delivered frames / (stage 1 + (3 + 3/38) windows + 99 pairs); VAE encode / decode of the enhancer
    (2 % of the job, profiles/r05_bench_full_pipeline.json) is not in the estimate -- `--workload full` measures the whole job."""
    import random
    from streamingt2v_amd import ops
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.enhance import DDIMSchedule, I2VEnhancer
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    chunk, H, W, cd = 38, 90, 160, 1024
    unet = I2VGenXLUNet(I2VConfig())
    unet.load_state_dict(init_by_name(unet.spec(), seed=5, device=device), device=device)
    g = torch.Generator(device=device); g.manual_seed(33)
    rn = lambda *sh: torch.randn(*sh, generator=g, device=device)
    il, emb, text = rn(1, 4, chunk, H, W) * 0.7, rn(1, cd), rn(1, 77, cd)
    conds = [dict(fps=torch.tensor([38, 38]), image_latents=torch.cat([il, il]), image_embeddings=torch.cat([torch.zeros_like(emb), emb]),
                  text=torch.cat([torch.zeros_like(text), text]))]
    video, noise = rn(1, 4, chunk, H, W) * 0.5, rn(1, 4, chunk, H, W)
    enh = I2VEnhancer(unet, DDIMSchedule(), guidance_scale=9.0, num_inference_steps=30, strength=0.97)
    n_steps = len(DDIMSchedule().get_timesteps(30, 0.97))
    with torch.no_grad():
        warm = I2VEnhancer(unet, DDIMSchedule(), guidance_scale=9.0, num_inference_steps=30, strength=0.1)      # 2 DDIM steps: one-time kernel attributes, caches
        warm.denoise(video, noise, conds, chunk, 0, rng=random.Random(33))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = enh.denoise(video, noise, conds, chunk, 0, rng=random.Random(33))
        torch.cuda.synchronize()
        win_s = time.perf_counter() - t0
        assert torch.isfinite(out).all()
        trace = LaunchTrace()
        ops.trace = trace
        warm.denoise(video, noise, conds, chunk, 0, rng=random.Random(33))          # traced separately (2 DDIM steps): the timed window above is untraced
        torch.cuda.synchronize()
        ops.trace = None
    eroof = roofline_from_trace(trace)
    eroof.pop("traced_kernels", None)
    eroof["traced"] = "2 DDIM steps of the window after the timed one; HIP events around every GEMM / fused-FF / MFMA-attention launch"
    del unet, enh, warm, out
    torch.cuda.empty_cache()
    vfi = EMAVFI(VFIConfig())
    vfi.load_state_dict(init_by_name(vfi.spec(), seed=3), device=device)
    fr = [torch.rand(720, 1280, 3, generator=g, device=device) for _ in range(4)]
    with torch.no_grad():
        for i in range(2):
            vfi.inference(fr[i], fr[i + 1], want_uint8=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_pairs = 8
        for i in range(n_pairs):
            vfi.inference(fr[i % 4], fr[(i + 1) % 4], want_uint8=True)
        torch.cuda.synchronize()
        pair_ms = (time.perf_counter() - t0) / n_pairs * 1e3
    del vfi
    torch.cuda.empty_cache()
    enh_s = win_s * (3.0 + 3.0 / 38.0)
    vfi_s = 99 * pair_ms * 1e-3
    return {"enhance_window_s": round(win_s, 3), "enhance_window": f"38 frames @ latent 90x160 (720x1280), {n_steps} DDIM steps, CFG 9 (batch 2 x 38)",
            "enhance_roofline": eroof, "vfi_pair_ms": round(pair_ms, 2), "vfi_pair": "EMA-VFI fast-TTA interpolation of one 720x1280 frame pair + uint8 conversion",
            "stage1_s_per_100_frames": round(stage1_s, 2), "enhance_s_estimate": round(enh_s, 2), "vfi_s_estimate": round(vfi_s, 2),
            "final_frames_per_s_estimate": round(180.0 / (stage1_s + enh_s + vfi_s), 4),
            "estimate_definition": "180 delivered frames / (stage-1 s per 100 frames + (3 + 3/38) x window s + 99 x pair s): the shipped job's frame arithmetic "
                                   "(3 blending windows of 38 frames + the 3-frame key-frame pre-pass, 99 interpolated pairs); the enhancer's VAE encode / decode is not counted "
                                   "(--workload full measures the whole job)"}


def run_stage1(args, rank, world, device):
    """Default workload: stage 1 of the 200-frame job, one step = one chunk of the real autoregressive sequence (see the module docstring).
    The warm-up ENDS at a video boundary (the stream starts (-warmup) mod 6 chunks into a video, on stand-in control frames), so the timed
    region always begins with chunk 0; every step is bracketed by a device synchronisation so that chunk 0 and AR chunks are timed apart.
    value = 100 frames / (mean chunk-0 time + 5 x mean AR-chunk time) of the timed steps: the whole-video rate, independent of where in the
    6-chunk cycle the K timed steps happen to stop (frames-kept / time, which depends on it, is reported next to it)."""
    from streamingt2v_amd import ops, parallel
    plan, model, c, uc, noises, g = build_stage1(args, world, device)
    start = (-args.warmup) % 6
    prev = (torch.rand(7, 3, 8 * LAT_H, 8 * LAT_W, generator=g, device=device) * 2 - 1) if start else None
    stream = Stage1Stream(model, c, uc, noises, start=start, prev_frames=prev)

    def sync_all():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        stream.step()
    sync_all()
    t0 = time.perf_counter()
    kept, t_prev, per_type = 0, t0, {0: [], 1: []}
    for _ in range(args.steps):
        k = stream.i % 6
        kept += stream.step()
        torch.cuda.synchronize()
        t_now = time.perf_counter()
        per_type[0 if k == 0 else 1].append(t_now - t_prev)
        t_prev = t_now
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    assert all(torch.isfinite(ch).all() for ch in stream.chunks)
    mean = lambda v: sum(v) / len(v) if v else None
    t_c0, t_ar = mean(per_type[0]), mean(per_type[1])
    if world > 1:                                    # slowest rank per chunk type, like the total
        t_c0 = parallel.max_over_ranks(t_c0, device=device) if t_c0 is not None else None
        t_ar = parallel.max_over_ranks(t_ar, device=device) if t_ar is not None else None
    kept_rate = plan.n_videos * kept / dt
    if t_c0 is not None and t_ar is not None:
        value, vdef = plan.n_videos * 100.0 / (t_c0 + 5.0 * t_ar), "100 frames / (mean chunk-0 s + 5 x mean AR-chunk s) of the timed steps"
    else:
        value, vdef = kept_rate, "frames kept / time (the timed steps hold only one chunk type: placement-dependent, use --steps >= 2)"
    roof = None
    if not args.no_trace and rank == 0 and world == 1:
        # separate traced pass: ONE AR chunk (the next chunk of the sequence if it is an AR chunk, else skip chunk 0 first)
        if stream.i % 6 == 0:
            stream.step()
        trace = LaunchTrace()
        ops.trace = trace
        stream.step()
        torch.cuda.synchronize()
        ops.trace = None
        roof = roofline_from_trace(trace)
    stages = None
    if not args.no_stages and rank == 0 and world == 1 and t_c0 is not None and t_ar is not None:
        try:
            stages = other_stages(args, device, t_c0 + 5.0 * t_ar)
        except Exception as e:       # the stage-1 measurement above must never be lost to a problem in the side measurement
            stages = {"error": repr(e)}
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline("stage1")
            except Exception as e:   # the GPU measurement above must never be lost to a host-side problem
                cpu = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"FAILED: {e!r}"}
        print(json.dumps({
            "metric": "generated frames/sec (576x1024)", "value": round(value, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": plan.scaling, "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": "StreamingSVD 200-frame job, stage 1 (100 frames @576x1024, BASELINE configs[2]): one step = one chunk of the "
                                   "autoregressive sequence chunk 0 (25 EDM steps) | AR1..AR5 (30 AYS steps, ControlNet(7 decoded frames) + CAM), "
                                   "CFG 2 x 25 frames @ latent 72x128 per evaluation, temporal-VAE decode of every chunk",
                       "value_definition": vdef, "chunk0_s_mean": None if t_c0 is None else round(t_c0, 4), "ar_chunk_s_mean": None if t_ar is None else round(t_ar, 4),
                       "chunks_timed": [len(per_type[0]), len(per_type[1])], "timed_region_starts_at": "chunk 0 (video boundary)",
                       "frames_kept_per_chunk": list(Stage1Stream.KEPT), "frames_in_timed_steps": kept, "kept_frames_over_time": round(kept_rate, 4),
                       "latent": [LAT_H, LAT_W], "denoise_steps": [args.denoise_steps or 25, args.denoise_steps or 30],
                       "parallelism": plan.describe(), "precision_plan": precision_plan(), "kernel_switches": kernel_switches(),
                       "weights": "seeded random, reference architecture (1.59 B + 0.67 B + 64 M parameters)"},
            "roofline": roof, "cpu_baseline": cpu, "stages": stages}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def run_full(args, rank, world, device):
    """--workload full (BASELINE configs[4], SURVEY 8d metric (ii)): one step = ONE whole 200-frame job, the script body of
    code/inference_i2v.py:227-259 -- image_to_video (stage 1: chunk 0 + 5 AR chunks, 100 frames @576x1024, uint8) -> enhance_video
    (frames resized to 720x1280, 2-D VAE encode, key-frame pre-pass, randomized blending: 3 windows of 38 frames, overlap 12, 29 DDIM
    steps, CFG 9, per-frame VAE decode) -> interpolate_video (EMA-VFI, fast TTA, dest_num_frames = 200).  value = final frames delivered
    (180: the reference drops the 10 frames that do not fill a blending window) / end-to-end seconds, host-side PIL resizes and device<->host copies between the stages included (they are part of the reference's job too).
    Random-weight networks at the reference's architectures; the CLIP TEXT tower is replaced by fixed random prompt embeddings (it runs
    once per job on two 77-token prompts).  --gpus N: stage 1 under the job plan, blending windows and VFI frame pairs sharded."""
    import numpy as np
    from streamingt2v_amd import parallel, pipeline as P
    from streamingt2v_amd.clip_vision import ClipVisionConfig, OpenCLIPVisionTower
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.enhance_codec import EnhanceCodec
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.temporal_ae import AutoencoderKL2D, VaeConfig
    plan, model, c, uc, noises, g = build_stage1(args, world, device)
    eunet = I2VGenXLUNet(I2VConfig())
    eunet.load_state_dict(init_by_name(eunet.spec(), seed=5, device=device), device=device)
    vae2d = AutoencoderKL2D(VaeConfig())
    vae2d.load_state_dict(init_by_name(vae2d.spec(), seed=21, device=device), device=device)
    tower = OpenCLIPVisionTower(ClipVisionConfig())
    tower.load_state_dict(init_by_name(tower.spec(), seed=22, device=device), device=device)
    vfi = EMAVFI(VFIConfig())
    vfi.load_state_dict(init_by_name(vfi.spec(), seed=3), device=device)
    torch.cuda.empty_cache()
    pipe = P.StreamingPipeline.__new__(P.StreamingPipeline)          # the stages are wired by hand: no checkpoint to load offline
    pipe.cfg = dict(P.DEFAULTS, enhance_steps=args.denoise_steps or P.DEFAULTS["enhance_steps"])
    pipe.model, pipe.enhancer_unet, pipe.vfi, pipe.device = model, eunet, vfi, device
    pipe.group = pipe.plan = None
    if world > 1 and plan.mode in ("job", "pairs"):
        pipe.group = plan.decode_group          # job: all ranks; pairs: the two ranks of this video
        if plan.mode == "job" and plan.sp is not None and plan.sp.size in (2, 4, 8):
            pipe.plan = plan                    # enhancer: CFG pair x frame<->pixel sequence parallelism (the 90x160 .. 12x20 levels split over 2 / 4 / 8 ranks)
    ge = torch.Generator(); ge.manual_seed(1)
    prompt = (torch.randn(1, 77, 1024, generator=ge), torch.randn(1, 77, 1024, generator=ge))
    image = (np.random.RandomState(7).rand(576, 1024, 3) * 255).astype("uint8")
    stage_s = {"stage1": 0.0, "enhance": 0.0, "vfi": 0.0}

    def one():
        with torch.no_grad():
            t = time.perf_counter()
            stream = Stage1Stream(model, c, uc, noises)
            for _ in range(6):
                stream.step()
            video = stream.video_u8.cpu().numpy()                                   # uint8 [100, 576, 1024, 3], like trainer.generated_video
            torch.cuda.synchronize(); t1 = time.perf_counter(); stage_s["stage1"] += t1 - t
            torch.manual_seed(33)                                                   # image latents sample from the global stream (pipeline_i2vgen_xl.py:486)
            codec = EnhanceCodec(vae2d, tower, generator=torch.Generator(device=device).manual_seed(8888), device=device)
            codec.set_prompt_embeds(*prompt)
            pipe.enhance_codec = codec
            video = pipe.enhance_video(image=image, video=video, use_randomized_blending=True, chunk_size=38, overlap_size=12)
            torch.cuda.synchronize(); t2 = time.perf_counter(); stage_s["enhance"] += t2 - t1
            out = pipe.interpolate_video(video, dest_num_frames=200)
            torch.cuda.synchronize(); stage_s["vfi"] += time.perf_counter() - t2
            return video.shape[0], out

    def sync_all():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    for k in stage_s:
        stage_s[k] = 0.0
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_enh, out = one()
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    # With randomized blending the reference keeps whole blending windows only (i2v_enhance_interface.py:109-113): 100 stage-1 frames ->
    # 3 windows of 38 with overlap 12 = 90 enhanced frames -> vfi_process(video_len=200) interpolates all 89 pairs and repeats the last frame
    # (:30-52) = 180 final frames.  value counts the frames the job actually delivers.
    n_final = out.shape[0]
    assert n_final == 2 * n_enh and out.shape[1:] == (720, 1280, 3) and str(out.dtype) == "uint8", (out.shape, out.dtype, n_enh)
    # roofline of the job's dominant kernel: ONE more job with HIP events around every GEMM / MFMA-attention launch of all three stages (the timed
    # region above is untraced); the GEMM family is one kernel template across stage 1, the enhancer, both VAEs and EMA-VFI.
    stage_mean = {k: round(v / args.steps, 2) for k, v in stage_s.items()}          # of the TIMED jobs (the traced job below would add to stage_s)
    roof = None
    if not args.no_trace and world == 1:
        from streamingt2v_amd import ops
        trace = LaunchTrace()
        ops.trace = trace
        try:
            one()
            torch.cuda.synchronize()
            ops.trace = None
            roof = roofline_from_trace(trace)
        except Exception as e:      # the timed measurement must not be lost to the trace pass
            ops.trace = None
            roof = {"error": repr(e)}
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline("full")
        except Exception as e:
            cpu = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"FAILED: {e!r}"}
    if rank == 0:
        print(json.dumps({
            "metric": "final frames/sec (200-frame job, 720x1280 output)", "value": round(plan.n_videos * args.steps * n_final / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": plan.scaling, "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": "full pipeline (BASELINE configs[4]; inference_i2v.py:227-259): stage 1 (100 frames @576x1024) + I2VGen-XL enhancement with "
                                   "randomized blending (3 windows x 38 frames, overlap 12, key-frame pre-pass, %d DDIM steps, CFG 9, 2-D VAE encode/decode @720x1280) "
                                   "+ EMA-VFI to 200 frames" % len(pipe_timesteps(pipe)),
                       "seconds_per_job": stage_mean, "frames_after_enhancement": int(n_enh),
                       "final_frames_per_job": int(n_final),
                       "parallelism": {"stage1": plan.describe(),
                                       "enhance": ("every window on all ranks: CFG pair x frame<->pixel sequence parallelism of degree %d inside I2VGenXLUNet (all-to-all around the "
                                                   "temporal layers, one all-gather of the two halves' predictions per window and DDIM step); key-frame pre-pass on the CFG pair" % plan.sp.size
                                                   if pipe.plan is not None else
                                                   "6 (window, CFG half) units + the key-frame pre-pass's 2 over the ranks of the video's group, one all-gather per DDIM step"
                                                   if pipe.group is not None else "single GPU"),
                                       "vfi": ("frame pairs sharded over the ranks of the video's group" if pipe.group is not None else "single GPU")},
                       "precision_plan": precision_plan(), "kernel_switches": kernel_switches(),
                       "weights": "seeded random, reference architectures (StreamingSVD 2.3 B, I2VGen-XL 1.42 B, AutoencoderKL, CLIP ViT-H/14 image tower, EMA-VFI 65.7 M); "
                                  "CLIP text tower replaced by fixed random prompt embeddings"},
            "roofline": roof, "cpu_baseline": cpu}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def pipe_timesteps(pipe):
    from streamingt2v_amd.enhance import DDIMSchedule
    return DDIMSchedule().get_timesteps(pipe.cfg["enhance_steps"], pipe.cfg["enhance_strength"])


def run_vfi(args, rank, world, device):
    """--workload vfi: one step = the frame interpolation of one 100-frame 720x1280 video to 200 frames
    (inference_i2v.py:252): 99 EMA-VFI inferences with fast TTA, each with its uint8 conversion; the frames are resident in HBM as
    fp32 BGR.  The PIL resize / host copies of vfi_process are not in the timed region."""
    from streamingt2v_amd import parallel
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.params import init_by_name
    model = EMAVFI(VFIConfig())
    model.load_state_dict(init_by_name(model.spec(), seed=3), device=device)
    g = torch.Generator(device=device); g.manual_seed(7 + rank)
    H, W, n_in = 720, 1280, 100
    video = [torch.rand(H, W, 3, generator=g, device=device) for _ in range(4)]          # 4 distinct frames cycled: content does not change the work

    def one():
        with torch.no_grad():
            return [model.inference(video[i % 4], video[(i + 1) % 4], want_uint8=True)[1] for i in range(n_in - 1)]

    def sync_all():
        torch.cuda.synchronize(); parallel.barrier(); torch.cuda.synchronize()

    for _ in range(args.warmup):
        one()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one()
    sync_all()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, device=device)
    assert len(out) == n_in - 1 and out[0].dtype == torch.uint8
    if rank == 0:
        print(json.dumps({
            "metric": "interpolated frames/sec (720x1280)", "value": round(world * args.steps * (n_in - 1) / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": "EMA-VFI (F=32, depth 2-2-2-4-4, fast TTA) 100 -> 200 frames @720x1280: 99 interpolations + uint8 conversion",
                       "frames_per_step": n_in - 1, "parallelism": f"replica-per-gpu x{world}", "weights": "seeded random, reference architecture (65.7 M parameters)"},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def self_spawn(n):
    """`python bench.py --gpus N ...` without a launcher: run the same command line as N ranks of ONE node under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1 at a free port) and pass its output / exit code through.  Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 6 = one whole video for stage1, 1 otherwise)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: 6 = one whole video for stage1, so that both chunk types are warmed; 1 otherwise)")
    ap.add_argument("--workload", default="stage1", choices=["stage1", "c2", "ar_chunk", "c3", "enhance", "vfi", "full"])
    ap.add_argument("--parallelism", default="auto", choices=["auto", "pairs", "job", "replica", "cfg"],
                    help="auto (default) = job for the stage1 / full workloads on an even number of GPUs (pairs for the chunk-level workloads), replica otherwise.  pairs: N/2 independent videos, each on a CFG pair of GPUs "
                         "(one 2-rank all-gather of the network output per Euler step; the only collectives of the default path); job (stage1 / full): ONE "
                         "job over all GPUs, CFG pair x frame<->pixel sequence parallelism (all-to-all around the temporal operators; strong scaling; "
                         "verified on the HIP kernels with 2 and 4 processes, tests/test_gpu_multiproc.py, never yet run over RCCL on 8 GPUs); replica: one "
                         "video per GPU (weak scaling, no collective); cfg (c2 / ar_chunk): GPU pairs split the CFG halves of one video")
    ap.add_argument("--denoise-steps", type=int, default=None, help="override Euler steps per chunk (debug only: INVALID for reporting)")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="16-bit element type of the kernels (fp32 accumulation either way, same MFMA rate): fp16 (default) = the reference's "
                         "own autocast precision (config.yaml:8), the one the parity tests assert north_star's tolerance in; bf16 selectable")
    ap.add_argument("--residual-stream", default=None, choices=["fp32", "16"],
                    help="residual stream of the UNet / ControlNet between kernels.  Default: the package's precision plan (round 4: fp32 in both networks + the "
                         "split-3 rim: StreamingWrapper.forward within north_star's 1e-3 of the reference on every frame); 16 = the 16-bit stream of rounds 2 / 3 "
                         "(~7 %% faster, 1.02e-3 mean / 1.20e-3 max)")
    ap.add_argument("--no-exact-rim", action="store_true", help="with --residual-stream 16: also run the condition embedding / stems / head / embedding MLPs "
                                                                "with plain 16-bit operands (the round-3 arithmetic: 1.15e-3 / 1.38e-3)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the per-step network evaluation from a hipGraph captured at the second Euler step of every chunk (sampling.EulerEDMSampler(use_graph=True))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="stage1 workload: skip the enhancer-window / VFI-pair side measurement after the timed region")
    ap.add_argument("--no-trace", action="store_true")
    args = ap.parse_args()
    if args.parallelism == "auto":
        # round 6: the metric is ONE 200-frame job at 1 / 2 / 4 / 8 GPUs, so the job-level workloads strong-scale one job by default (CFG pair x sequence
        # parallelism; at N = 2 that IS one CFG pair) -- `--parallelism pairs` is the throughput plan (N/2 independent videos, weak scaling beyond 2 GPUs)
        if args.gpus > 1 and args.gpus % 2 == 0:
            args.parallelism = "job" if args.workload in ("stage1", "full") else "pairs"
        else:
            args.parallelism = "replica"
    if args.steps is None:
        args.steps = 6 if args.workload == "stage1" else 1
    if args.warmup is None:
        args.warmup = {"stage1": 6, "full": 0}.get(args.workload, 1)          # full: one job is 100 s; its stage 1 warms the kernels

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        return self_spawn(args.gpus)        # `python bench.py --gpus N`: re-exec as N ranks under torch.distributed.run
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # SVD_BENCH_SHARE_GPU=1 (tests on a 1-GPU box): every rank uses cuda:0 and the collectives go over gloo with host staging
    share = os.environ.get("SVD_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    from streamingt2v_amd import ops, parallel
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.streaming_svd import StreamingSVD
    parallel.init_from_env(backend="gloo" if share else "nccl", device=device)
    ops.set_element_dtype(torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    if args.residual_stream == "fp32":
        ops.set_stream_f32(True)
    elif args.residual_stream == "16":
        ops.set_stream_f32(False)
        ops.set_precision_plan(exact_rim=not args.no_exact_rim, cn_stream_f32=not args.no_exact_rim, stream_f32_min_ch=0)
    if args.workload == "stage1":
        return run_stage1(args, rank, world, device)
    if args.workload == "full":
        return run_full(args, rank, world, device)
    if args.workload == "enhance":
        return run_enhance(args, rank, world, device)
    if args.workload == "c3":
        return run_c3(args, rank, world, device)
    if args.workload == "vfi":
        return run_vfi(args, rank, world, device)

    # parallelism: "replica" = one independent video per rank (no data-path collective);
    #              "cfg"     = ranks (2k, 2k+1) split the two CFG halves of one video, one RCCL all-gather per Euler step
    cfg_ex, n_videos, video_id = None, world, rank
    if args.parallelism in ("job", "pairs"):
        args.parallelism = "replica"            # c2 / ar_chunk are single-chunk parity workloads: replicas unless --parallelism cfg
    if args.parallelism == "cfg":
        import torch.distributed as dist
        assert world % 2 == 0, "--parallelism cfg needs an even number of GPUs"
        groups = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
        cfg_ex = parallel.CfgPairExchange(groups[rank // 2])
        n_videos, video_id = world // 2, rank // 2

    wrapper, vae = build_models(args.workload, device)
    n_steps = args.denoise_steps or (25 if args.workload == "c2" else 30)
    min_scale = 1.0 if args.workload == "c2" else 1.5      # diffusers SVD default vs config.yaml:153-156
    from streamingt2v_amd.sampling import AlignYourSteps, EDMDiscretization
    disc = EDMDiscretization() if args.workload == "c2" else AlignYourSteps()     # chunk 0: Karras/EDM ; AR chunks: AYS
    sampler = EulerEDMSampler(num_steps=n_steps, num_frames=T_FRAMES, min_scale=min_scale, max_scale=3.0, discretization=disc,
                              cfg_exchange=cfg_ex)
    model = StreamingSVD(wrapper, vae, sampler)
    c, uc, ctrl, noise = synthetic_inputs(device, 33 + video_id)

    def one_chunk():
        with torch.no_grad():
            return model._generate_conditional_output(c, uc, ctrl if args.workload == "ar_chunk" else None, noise)

    def sync_all():
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_chunk()
    trace = None if args.no_trace else LaunchTrace()
    ops.trace = trace
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frames = one_chunk()
    sync_all()
    dt = time.perf_counter() - t0
    ops.trace = None
    assert torch.isfinite(frames).all()
    dt = parallel.max_over_ranks(dt, device=device)

    if rank == 0:
        roof = roofline_from_trace(trace) if trace is not None else None
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            try:
                cpu = cpu_baseline(args.workload)
            except Exception as e:   # the GPU measurement above must never be lost to a host-side problem
                cpu = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"FAILED: {e!r}"}
        out = {
            "metric": "generated frames/sec (576x1024)", "value": round(n_videos * args.steps * T_FRAMES / dt, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak" if args.parallelism == "replica" else "strong", "vs_baseline": None, "dtype": args.dtype.replace("fp16", "f16"), "data": "synthetic",
            "config": {"workload": ("SVD-XT UNet single chunk: 25 frames @576x1024, CFG 2, %d Euler-EDM steps (Karras rho=7 schedule, guidance 1->3) + temporal-VAE decode"
                                    if args.workload == "c2" else
                                    "StreamingSVD AR chunk: 25 frames @576x1024, CFG 2, %d AYS steps, ControlNet(7)+CAM + temporal-VAE decode") % n_steps,
                       "frames_per_step": T_FRAMES, "denoise_steps": n_steps, "latent": [LAT_H, LAT_W],
                       "parallelism": (f"replica-per-gpu x{world}" if args.parallelism == "replica" else f"cfg-pair split x{world // 2} videos (RCCL all-gather per step)"), "weights": "seeded random, reference architecture"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
