#!/usr/bin/env python
"""Microbenchmark of svd_gemm over (shape, epilogue, tile config) -- used to separate per-tile fixed cost from the
K loop (run on an MI355X)."""
import sys, os, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize()
        best = min(best, s.elapsed_time(e))
    return best

def main():
    dev = "cuda"
    M = 460800
    cfgs = [1, 2, 5, 8, 13, 3, 12]
    print("M=%d; columns: cfg -> ms (TFLOP/s)" % M)
    for (N, K, geglu, res) in [(2560, 320, False, False), (2560, 320, True, False), (2560, 640, True, False), (2560, 1280, True, False),
                               (320, 320, False, False), (320, 320, False, True), (320, 1280, False, True), (640, 320, False, False),
                               (1280, 320, False, False), (960, 320, False, False)]:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev)
        R = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
        out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.bfloat16)
        line = f"N={N:5d} K={K:5d} geglu={int(geglu)} res={int(res)} :"
        for c in cfgs:
            try:
                ms = t(lambda: ops.gemm(a, w, bias=bias, geglu=geglu, residual=R, out=out, tile_cfg=c))
                line += f"  {c}:{ms:.3f}({2.0*M*N*K/ms/1e9:.0f})"
            except Exception as e:
                line += f"  {c}:--"
        print(line, flush=True)
        del a, w, out, R

if __name__ == "__main__":
    main()
