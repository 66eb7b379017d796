"""A/B of the XCD rendezvous of the panel-walk GEMM launches (probe library libsvdhip_pv_rdv.so, SVD_GEMM_RDV=0|1 read per launch): one process, interleaved, best of three.
    SVD_LIB_FILE=libsvdhip_pv_rdv.so python tools/gemm_rdv_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402

dt = torch.float16
ops.set_element_dtype(dt)
g = torch.Generator(device="cuda")
g.manual_seed(0)
CASES = [("stage 1 level-1 GEGLU proj", 115200, 5120, 640, True), ("stage 1 level-2 GEGLU proj", 28800, 10240, 1280, True), ("stage 1 level-3 GEGLU proj", 7200, 10240, 1280, True),
         ("stage 1 level-2 temporal q|k|v", 28800, 3840, 1280, False), ("enhancer level-1 GEGLU proj", 273600, 5120, 640, True), ("enhancer level-2 GEGLU proj", 68400, 10240, 1280, True)]


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'case':34s} {'M':>7s} {'N':>6s} {'K':>5s}   no rendezvous      rendezvous")
for name, M, N, K, geglu in CASES:
    a = torch.randn(M, K, generator=g, device="cuda").to(dt)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(dt)
    bias = torch.randn(N, generator=g, device="cuda")
    run = lambda: ops.gemm(a, w, bias=bias, geglu=geglu)
    best, outs = {}, {}
    for rep in range(3):
        for v in ("0", "1"):
            os.environ["SVD_GEMM_RDV"] = v
            best[v] = min(best.get(v, 1e30), timed(run))
            if rep == 0:
                outs[v] = run().clone()
    os.environ.pop("SVD_GEMM_RDV", None)
    fl = 2.0 * M * N * K
    print(f"{name:34s} {M:7d} {N:6d} {K:5d} {best['0']:8.1f} us {fl / best['0'] / 1e6:5.0f} TF {best['1']:8.1f} us {fl / best['1'] / 1e6:5.0f} TF   bit-identical {torch.equal(outs['0'], outs['1'])}", flush=True)
