"""CPU suite, part 4: the N>1 paths with torch.distributed gloo, world_size 2 (no GPU).

  * randomized blending: single-process product loop == oracle restatement (bit-exact) and the window-sharded
    all-gather version == single process, on 2 ranks;
  * CFG-pair exchange: each rank evaluates one CFG half, all-gather == the 2-batch evaluation;
  * EMA-VFI frame pairs sharded over ranks (all-gather of the middle frames) == single process;
  * bench timing helper: max over ranks."""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _denoise(idx, w):                      # deterministic stand-in for UNet + CFG + DDIM step of a window
    return w * (1.0 + 0.125 * idx) + 0.5 * idx


def _case():
    g = torch.Generator(); g.manual_seed(7)
    return torch.randn(1, 4, 90, 3, 5, generator=g), 38, 12, 3          # shipped: 38 / 12, 3 windows over 90 frames


def test_blending_matches_oracle_and_reference_quirks():
    from oracle.blending_oracle import randomized_blending_step
    from streamingt2v_amd.blending import blend_step, chunk_starts
    lat, cs, ov, n = _case()
    a = blend_step(lat, _denoise, cs, ov, n, random.Random(33))
    b = randomized_blending_step(lat, _denoise, cs, ov, n, random.Random(33))
    assert torch.equal(a, b)
    assert chunk_starts(90, 38, 12, 3) == [0, 26, 52]
    with pytest.raises(NotImplementedError):                           # 100 frames do not divide
        blend_step(torch.zeros(1, 4, 88, 2, 2), _denoise, cs, ov, n, random.Random(1))
    with pytest.raises(NotImplementedError):
        randomized_blending_step(torch.zeros(1, 4, 88, 2, 2), _denoise, cs, ov, n, random.Random(1))
    # overlap 0 draws no random numbers (pipeline_i2vgen_xl.py:894-898)
    r = random.Random(5); st = r.getstate()
    blend_step(torch.zeros(1, 4, 20, 2, 2), _denoise, 10, 0, 2, r)
    assert r.getstate() == st


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from streamingt2v_amd import parallel
    from streamingt2v_amd.blending import blend_step, blend_step_sharded
    assert parallel.init_from_env(backend="gloo") == world
    try:
        lat, cs, ov, n = _case()
        ref = blend_step(lat, _denoise, cs, ov, n, random.Random(33))
        got = blend_step_sharded(lat, _denoise, cs, ov, n, random.Random(33))
        ok_blend = torch.equal(ref, got)
        # 5 windows on 2 ranks: uneven sharding with a padded slot
        lat5 = torch.arange(1 * 2 * 46 * 2 * 2, dtype=torch.float32).reshape(1, 2, 46, 2, 2)
        ok_blend5 = torch.equal(blend_step(lat5, _denoise, 14, 6, 5, random.Random(3)),
                                blend_step_sharded(lat5, _denoise, 14, 6, 5, random.Random(3)))
        # CFG pair: net(x2, cond2) evaluated half per rank
        g = torch.Generator(); g.manual_seed(1)
        x = torch.randn(6, 4, generator=g)
        cond = torch.stack([torch.zeros(6, 4), torch.ones(6, 4)])       # [uncond | cond]
        net = lambda xx, cc: xx * 2.0 + cc * 3.0
        full = torch.cat([net(x, cond[0]), net(x, cond[1])], 0)
        ex = parallel.CfgPairExchange()
        ok_cfg = torch.equal(ex.gather(net(x, cond[ex.half])), full)
        # EMA-VFI: frame pairs sharded over the ranks, one all-gather of the uint8 middle frames == the single-process video
        import numpy as np
        from streamingt2v_amd.ema_vfi import vfi_process

        class FakeVFI:                      # stands in for EMAVFI.inference (a GPU kernel path): any deterministic function of the pair
            def inference(self, a, b, want_uint8=False):
                m = (a * 0.25 + b * 0.75)
                return m, (m * 255.0).to(torch.uint8)

        rs = np.random.default_rng(11)
        video = [rs.integers(0, 256, (8, 12, 3), dtype=np.uint8) for _ in range(6)]
        ok_vfi = True
        for n in (11, 12, 4):               # 5 pairs on 2 ranks (padded slot), even target length, 1 pair (rank 1 idle)
            one = vfi_process(video[: (n + 1) // 2], FakeVFI(), n, out_size=(12, 8), device="cpu")
            two = vfi_process(video[: (n + 1) // 2], FakeVFI(), n, out_size=(12, 8), device="cpu", sharded=True)
            ok_vfi = ok_vfi and len(one) == len(two) == n and all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(one, two))
        tmax = parallel.max_over_ranks(1.0 + rank)
        parallel.barrier()
        out.put((rank, ok_blend and ok_vfi, ok_blend5, ok_cfg, tmax, parallel.shard_items(5, rank, world)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_paths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_blend, ok_blend5, ok_cfg, tmax, items in res:
        assert ok_blend and ok_blend5 and ok_cfg
        assert tmax == 2.0
        assert items == ([0, 2, 4] if rank == 0 else [1, 3])


# ---- sequence parallelism of the denoiser forward (SURVEY.md 8e option 2) + CFG pair + sharded decode, on gloo ranks -------------------
def _sp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.set_num_threads(2)
    from tests import svd_shim
    svd_shim.install()                                   # fp32 torch statements of the HIP launchers: the HOST logic is what runs here
    from oracle import cases
    from streamingt2v_amd import parallel
    from streamingt2v_amd.params import init_by_name
    from streamingt2v_amd.sampling import EulerEDMSampler
    from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
    from streamingt2v_amd.wrappers import StreamingWrapper
    assert parallel.init_from_env(backend="gloo") == world
    try:
        torch.set_grad_enabled(False)
        tu = cases.TINY_UNET
        cfg = UNetConfig(num_res_blocks=tu["num_res_blocks"], attention_resolutions=tu["attention_resolutions"], channel_mult=tu["channel_mult"],
                         conditioning_embedding_out_channels=tu["cond_embed"])
        unet, cn = VideoUNet(cfg), ControlNet(cfg)
        unet.load_state_dict(init_by_name(unet.spec(), seed=1), device="cpu")
        cn.load_state_dict(init_by_name(cn.spec(), seed=2), device="cpu")
        T, Tc = tu["T"], tu["Tc"]
        inp = cases.tiny_wrapper_inputs()
        c = {k: inp[k] for k in ("concat", "crossattn", "vector")}
        kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T), ctrl_frames=inp["ctrl_frames"])
        wrap = StreamingWrapper(unet, cn, Tc)
        ref = wrap.forward(inp["x"], inp["t"], c, **kw)                       # single process: every rank computes the same reference
        # (1) sequence parallelism alone (degree 2: the tiny case has Tc = 3 conditioning frames), CFG batch 2 (B = 2: exercises the batch
        #     interleave of the all-to-alls); with 4 ranks: two independent SP groups {0,1} and {2,3}
        sp_groups = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
        wrap.sp = parallel.SeqParallel(sp_groups[rank // 2])
        got = wrap.forward(inp["x"], inp["t"], c, **kw)
        e_sp = (got - ref).abs().max().item()
        # also without control frames (chunk 0)
        kw0 = dict(kw, ctrl_frames=None)
        e_sp0 = (wrap.forward(inp["x"], inp["t"], c, **kw0) - StreamingWrapper(unet, cn, Tc).forward(inp["x"], inp["t"], c, **kw0)).abs().max().item()
        wrap.sp = None
        # (2) the job plan of bench.py: CFG pair x SP(world / 2) through the fused sampler, 2 Euler steps
        sin = cases.tiny_sampler_inputs()
        z_ref = EulerEDMSampler(num_steps=2, num_frames=T)(wrap, sin["noise"].clone(), sin["c"], sin["uc"], batch_size=2, num_video_frames=T,
                                                           ctrl_frames=inp["ctrl_frames"])
        plan = parallel.JobPlan(world, rank, "job")

        class _Vae:
            decode_group = None
        vae = _Vae()
        plan.attach(wrap, vae)
        z = EulerEDMSampler(num_steps=2, num_frames=T, cfg_exchange=plan.cfg_exchange)(wrap, sin["noise"].clone(), sin["c"], sin["uc"], batch_size=2,
                                                                                       num_video_frames=T, ctrl_frames=inp["ctrl_frames"])
        e_job = ((z - z_ref).abs().max() / z_ref.abs().max()).item()
        # (3) decode groups sharded over the ranks == local decode (stand-in decoder: any function of the group's latents)
        from streamingt2v_amd.streaming_svd import StreamingSVD

        class _Dec:
            decode_group = None
            def decode(self, zc, timesteps=None, clamp=False):
                return (zc.mean(1, keepdim=True).repeat(1, 3, 1, 1) * timesteps).repeat_interleave(8, 2).repeat_interleave(8, 3).contiguous()
            @staticmethod
            def output_shape(zz):
                return (zz.shape[0], 3, 8 * zz.shape[2], 8 * zz.shape[3])
        dec = _Dec()
        zz = torch.randn(11, 4, 2, 3, generator=torch.Generator().manual_seed(3))      # groups of 8 + 3
        one = StreamingSVD(None, dec).decode_first_stage(zz)
        dec.decode_group = vae.decode_group
        two = StreamingSVD(None, dec).decode_first_stage(zz)
        # (4) bench.py's DEFAULT plan: world / 2 independent videos, each on a CFG pair (no sequence parallelism); decode split inside the pair
        wrap.sp = None
        pplan = parallel.JobPlan(world, rank, "pairs")
        pvae = _Vae()
        pplan.attach(wrap, pvae)
        zp = EulerEDMSampler(num_steps=2, num_frames=T, cfg_exchange=pplan.cfg_exchange)(wrap, sin["noise"].clone(), sin["c"], sin["uc"], batch_size=2,
                                                                                        num_video_frames=T, ctrl_frames=inp["ctrl_frames"])
        e_pairs = ((zp - z_ref).abs().max() / z_ref.abs().max()).item()
        dec.decode_group = pvae.decode_group
        three = StreamingSVD(None, dec).decode_first_stage(zz)
        pairs_ok = (pplan.mode == "pairs" and pplan.n_videos == world // 2 and pplan.video_id == rank // 2 and torch.equal(one, three)
                    and pplan.scaling == ("strong" if world == 2 else "weak") and wrap.sp is None)
        out.put((rank, e_sp, e_sp0, e_job, torch.equal(one, two), plan.describe(), plan.scaling, e_pairs, pairs_ok))
    finally:
        dist.destroy_process_group()


# ---- enhancement stage: (window, CFG half) units sharded over the ranks (blending.blend_step_units_sharded) ---------------------------------
def _ph(idx, half, w):                      # stand-in for one CFG half of the UNet evaluation of a window: [B, C, chunk, H, W] -> [chunk, C, H, W]
    return (w[0].permute(1, 0, 2, 3) * (1.0 + 0.25 * half) + 0.125 * idx).contiguous()


def _cb(idx, w, pu, pc):                    # stand-in for guidance + scheduler step
    return (w[0].permute(1, 0, 2, 3) - 0.5 * (pu + 9.0 * (pc - pu))).permute(1, 0, 2, 3)[None]


def _units_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from streamingt2v_amd import parallel
    from streamingt2v_amd.blending import blend_step, blend_step_units_sharded
    assert parallel.init_from_env(backend="gloo") == world
    try:
        lat, cs, ov, n = _case()
        ok = True
        for nn, ll, c, o in ((n, lat, cs, ov), (1, lat[:, :, :38], 38, 0), (5, torch.arange(1 * 2 * 46 * 2 * 2, dtype=torch.float32).reshape(1, 2, 46, 2, 2), 14, 6)):
            ref = blend_step(ll, lambda idx, w: _cb(idx, w, _ph(idx, 0, w), _ph(idx, 1, w)), c, o, nn, random.Random(33))
            got = blend_step_units_sharded(ll, _ph, _cb, c, o, nn, random.Random(33))
            ok = ok and torch.equal(ref, got)
        parallel.barrier()
        out.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_enhancer_units_sharded_equals_single_process(world):
    """3 windows x 2 CFG halves = 6 units on 2 / 3 / 8 ranks (8: two ranks hold no unit and still join the all-gather); the 1-window key-frame
    pre-pass (2 units) and a 5-window case with padded slots: bit-identical latents on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_units_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _ in res] == list(range(world)) and all(ok for _, ok in res), res


@pytest.mark.parametrize("world", [2, 4])
def test_sequence_parallel_forward_equals_single_process(world):
    """The frame <-> pixel sequence-parallel StreamingWrapper forward (all-to-all around the temporal operators, all-reduced 5-D
    GroupNorm sums, all-gathered CAM keys / values and network output) reproduces the single-process forward on every rank; so does
    the CFG-pair x SP job plan through the fused sampler; the sharded decode is bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, e_sp, e_sp0, e_job, dec_ok, desc, scaling, e_pairs, pairs_ok in res:
        assert e_pairs < 2e-4 and pairs_ok, (rank, e_pairs, pairs_ok)
        assert e_sp < 2e-4 and e_sp0 < 2e-4, (rank, e_sp, e_sp0)          # fp32 on both sides: summation order only
        assert e_job < 2e-4, (rank, e_job)
        assert dec_ok and scaling == "strong"
        assert ("sequence parallelism of degree %d" % (world // 2)) in desc or world == 2


# ---- a data-path backend that HANGS inside a collective: the preflight times out, the plan falls back to replicas, the bench's barrier moves ----

def _hang_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SVD_PREFLIGHT_TIMEOUT="4")
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from streamingt2v_amd import parallel
    assert parallel.init_from_env(backend="gloo") == world
    real = parallel.CfgPairExchange.gather
    if rank == 1:                                       # rank 1 never enters the first collective of the plan: rank 0 blocks inside it for good
        parallel.CfgPairExchange.gather = lambda self, x: time.sleep(3600)
    t0 = time.perf_counter()
    plan = parallel.JobPlan(world, rank, "pairs")
    dt = time.perf_counter() - t0
    parallel.CfgPairExchange.gather = real
    parallel.barrier()                                  # must not touch the hung default group
    slowest = parallel.max_over_ranks(float(rank + 1))
    assert plan.abort_report is not None and "host-side" in plan.abort_report          # no device here: nothing to abort, and said so
    out.put((rank, plan.mode, plan.fallback_reason, plan.n_videos, plan.video_id, parallel._CONTROL["use"], slowest, dt))
    out.close(); out.join_thread()
    os._exit(0)                                         # the helper threads are parked inside the dead collective: no orderly teardown


def test_preflight_times_out_on_a_hung_backend_and_falls_back_to_replicas():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hang_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank, mode, why, n_videos, video_id, use_ctl, slowest, dt in res:
        assert mode == "replica" and n_videos == 2 and video_id == rank, (rank, mode)
        assert why is not None and "preflight failed" in why and use_ctl, (rank, why)
        assert slowest == 2.0 and dt < 60.0, (rank, slowest, dt)
    assert any("timed out" in r[2] for r in res)


# ---- enhancement stage, round 5: frame <-> pixel sequence parallelism inside I2VGenXLUNet, and the CFG-pair x SP plan of I2VEnhancer.denoise -------

def _enh_sp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.set_num_threads(2)
    from tests import svd_shim
    svd_shim.install()
    from oracle import cases
    from streamingt2v_amd import parallel
    from streamingt2v_amd.enhance import DDIMSchedule, I2VEnhancer
    from streamingt2v_amd.i2vgen_unet import I2VConfig, I2VGenXLUNet
    from streamingt2v_amd.params import init_by_name
    assert parallel.init_from_env(backend="gloo") == world
    try:
        torch.set_grad_enabled(False)
        kw, ti = cases.tiny_i2v_kwargs(), cases.TINY_I2V
        unet = I2VGenXLUNet(I2VConfig(block_out_channels=kw["block_out_channels"], layers_per_block=kw["layers_per_block"],
                                      cross_attention_dim=kw["cross_attention_dim"], attn_levels=(True, True, False)))
        unet.load_state_dict(init_by_name(unet.spec(), seed=5), device="cpu")
        inp = cases.tiny_i2v_inputs()
        call = lambda: unet(inp["sample"], inp["t"], fps=inp["fps"], image_latents=inp["image_latents"], image_embeddings=inp["image_embeddings"],
                            encoder_hidden_states=inp["text"])[0]
        ref = call()                                                       # CFG batch 2 x 6 frames @ 9 x 16: single process
        # (1) the UNet alone, sequence parallel over groups of 2 ranks (4 ranks: two independent groups); uneven frames with degree 4 (6 = 2 2 1 1)
        groups = [dist.new_group([2 * k, 2 * k + 1]) for k in range(world // 2)]
        unet.sp = parallel.SeqParallel(groups[rank // 2])
        e_sp2 = (call() - ref).abs().max().item()
        e_sp4 = 0.0
        if world == 4:
            unet.sp = parallel.SeqParallel(dist.group.WORLD)
            e_sp4 = (call() - ref).abs().max().item()
        unet.sp = None
        # (2) the denoising loop with randomized blending on the stage-1 job plan (CFG pair x SP of degree world / 2) against the single-process loop:
        #     2 overlapping windows of 6 frames, 3 DDIM steps; then a 1-window "key-frame pre-pass" of 1 frame (fewer frames than ranks: the pair alone)
        chunk, overlap, H, W, cd = ti["F"], 2, ti["h"], ti["w"], ti["cross_attention_dim"]
        n_frames = 2 * chunk - overlap
        g = torch.Generator(); g.manual_seed(2024)
        video, noise = torch.randn(1, 4, n_frames, H, W, generator=g) * 0.5, torch.randn(1, 4, n_frames, H, W, generator=g)
        conds = []
        for i in range(2):
            il = torch.randn(1, 4, chunk, H, W, generator=g) * 0.7
            emb, text = torch.randn(1, cd, generator=g), torch.randn(1, ti["text_tokens"], cd, generator=g)
            conds.append(dict(fps=torch.tensor([8, 8]), image_latents=torch.cat([il, il]), image_embeddings=torch.cat([torch.zeros_like(emb), emb]),
                              text=torch.cat([torch.zeros_like(text), text])))
        enh = I2VEnhancer(unet, DDIMSchedule(), guidance_scale=9.0, num_inference_steps=10, strength=0.35)
        one = enh.denoise(video, noise, conds, chunk, overlap, rng=random.Random(33))
        plan = parallel.JobPlan(world, rank, "job", frames_cond=chunk, min_pix=12)
        two = enh.denoise(video, noise, conds, chunk, overlap, rng=random.Random(33), plan=plan)
        e_loop = ((two - one).abs().max() / one.abs().max()).item()
        c1 = [dict(conds[0], image_latents=conds[0]["image_latents"][:, :, :1])]
        k_one = enh.denoise(video[:, :, :1], noise[:, :, :1], c1, 1, 0, rng=random.Random(33))
        k_two = enh.denoise(video[:, :, :1], noise[:, :, :1], c1, 1, 0, rng=random.Random(33), plan=plan)
        e_key = ((k_two - k_one).abs().max() / k_one.abs().max()).item()
        out.put((rank, e_sp2, e_sp4, e_loop, e_key, plan.mode, unet.sp is None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_enhancer_sequence_parallel_equals_single_process(world):
    """The frame <-> pixel split of the enhancer (TransformerTemporalModel / TemporalConvLayer in the pixel layout with all-reduced GroupNorm sums,
    everything else on the rank's frames, all-gather of the prediction) reproduces I2VGenXLUNet.forward on every rank; I2VEnhancer.denoise on the
    stage-1 job plan (CFG pair x sequence parallelism) reproduces the single-process loop with randomized blending, pre-pass included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_enh_sp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, e_sp2, e_sp4, e_loop, e_key, mode, sp_restored in res:
        assert e_sp2 < 2e-4 and e_sp4 < 2e-4, (rank, e_sp2, e_sp4)          # fp32 on both sides: summation order of the pooled norms only
        assert e_loop < 5e-4 and e_key < 5e-4, (rank, e_loop, e_key)
        assert mode == "job" and sp_restored
