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
