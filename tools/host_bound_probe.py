"""Is the job host-bound?  One AR chunk (30 AYS steps, ControlNet + CAM, decode): wall time until the host has ENQUEUED everything (the call
returns) vs until the GPU has finished, and the number of kernel launches the libsvdhip entry points made.   python tools/host_bound_probe.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from streamingt2v_amd import ops, lib as L
from streamingt2v_amd.sampling import AlignYourSteps, EulerEDMSampler
from streamingt2v_amd.streaming_svd import StreamingSVD

torch.cuda.set_device(0)
ops.set_element_dtype(torch.float16)
wrapper, vae = bench.build_models("ar_chunk", "cuda:0")
sampler = EulerEDMSampler(num_steps=30, num_frames=25, min_scale=1.5, max_scale=3.0, discretization=AlignYourSteps())
model = StreamingSVD(wrapper, vae, sampler)
c, uc, ctrl, noise = bench.synthetic_inputs("cuda:0", 33)
calls = {"n": 0}
orig = ops.check
def counting_check(rc, what):
    calls["n"] += 1
    return orig(rc, what)
ops.check = counting_check
with torch.no_grad():
    model._generate_conditional_output(c, uc, ctrl, noise); torch.cuda.synchronize()
    for rep in range(2):
        calls["n"] = 0
        t0 = time.perf_counter()
        model._generate_conditional_output(c, uc, ctrl, noise)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"AR chunk: host returned after {t1 - t0:.3f} s, GPU finished after {t2 - t0:.3f} s; {calls['n']} libsvdhip calls "
              f"({(t1 - t0) / max(calls['n'], 1) * 1e6:.1f} us of host time per call)", flush=True)

# ---- part 1b: the same AR chunk with the per-step forward replayed from a hipGraph (captured at step 1 of the chunk) ---------------------------
gs = EulerEDMSampler(num_steps=30, num_frames=25, min_scale=1.5, max_scale=3.0, discretization=AlignYourSteps(), use_graph=True)
gmodel = StreamingSVD(wrapper, vae, gs)
with torch.no_grad():
    ref = model._generate_conditional_output(c, uc, ctrl, noise)
    got = gmodel._generate_conditional_output(c, uc, ctrl, noise); torch.cuda.synchronize()
    print(f"hipGraph replay of the per-step forward: frames bit-identical to the eager chunk: {bool(torch.equal(ref, got))}", flush=True)
    for rep in range(2):
        calls["n"] = 0
        t0 = time.perf_counter()
        gmodel._generate_conditional_output(c, uc, ctrl, noise)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"AR chunk, graphed steps: host returned after {t1 - t0:.3f} s, GPU finished after {t2 - t0:.3f} s; {calls['n']} libsvdhip calls from Python "
              f"(step 0 eager + one capture + decode)", flush=True)
del gmodel, gs

# ---- part 2: host cost of a launch WITHOUT queue back-pressure -------------------------------------------------------------------------
# The shipped architecture on a 16x16 latent (kernels of a few microseconds: the GPU drains the queue faster than Python fills it), so the
# time until forward() returns is pure host work: Python layer logic + ctypes marshalling + hipLaunchKernel.  This is the number that bounds
# a sequence-parallel rank (1/S of the GPU work behind the SAME number of launches).
del model, wrapper, vae
torch.cuda.empty_cache()
from oracle.cases import FULLARCH_CASE as fc, fullarch_inputs
from streamingt2v_amd.params import init_by_name
from streamingt2v_amd.video_model import ControlNet, UNetConfig, VideoUNet
from streamingt2v_amd.wrappers import StreamingWrapper
cfg = UNetConfig()
unet, cn = VideoUNet(cfg), ControlNet(cfg)
unet.load_state_dict(init_by_name(unet.spec(), seed=33, device="cuda:0"), device="cuda:0")
cn.load_state_dict(init_by_name(cn.spec(), seed=34, device="cuda:0"), device="cuda:0")
wrap = StreamingWrapper(unet, cn, fc["Tc"])
inp = {k: v.cuda() for k, v in fullarch_inputs().items()}
T = fc["T"]
kw = dict(batch_size=2, num_video_frames=T, image_only_indicator=torch.zeros(2, T, device="cuda"), ctrl_frames=inp["ctrl_frames"])
cc = {k: inp[k] for k in ("concat", "crossattn", "vector")}
with torch.no_grad():
    for _ in range(3):
        wrap.forward(inp["x"], inp["t"], cc, **kw)
    torch.cuda.synchronize()
    calls["n"] = 0
    t0 = time.perf_counter()
    for _ in range(5):
        wrap.forward(inp["x"], inp["t"], cc, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
n = calls["n"] / 5
print(f"shipped architecture on a {fc['h']}x{fc['w']} latent ({2 * T} frames): {n:.0f} libsvdhip calls per forward; host returns after {(t1 - t0) / 5 * 1e3:.1f} ms "
      f"({(t1 - t0) / 5 / n * 1e6:.1f} us of host time per call), GPU done after {(t2 - t0) / 5 * 1e3:.1f} ms  -> a forward costs >= {(t1 - t0) / 5 * 1e3:.0f} ms of "
      f"host time whatever the GPU share of a rank is")

# ---- part 3: host cost of a REPLAYED forward (hipGraph) on the same small case ------------------------------------------------------------------
with torch.no_grad():
    x = inp["x"][:T].contiguous()
    c2 = {k: cc[k].float().contiguous() for k in ("vector", "crossattn", "concat")}
    scale, tvec = wrap.step_scalars(x, 2)
    scale.fill_(0.5); tvec.fill_(0.1)
    kwg = dict(batch_size=2, num_video_frames=T, ctrl_frames=inp["ctrl_frames"])
    ref = wrap.forward_fused_static(x, c2, **kwg)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = wrap.forward_fused_static(x, c2, **kwg)
    g.replay(); torch.cuda.synchronize()
    same = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"same forward replayed from a hipGraph (bit-identical: {same}): host returns after {(t1 - t0) / 5 * 1e3:.2f} ms per forward, GPU done after "
      f"{(t2 - t0) / 5 * 1e3:.1f} ms  -> the host cost of a step no longer depends on the number of kernels")
