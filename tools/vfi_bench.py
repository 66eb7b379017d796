"""Time the native EMA-VFI at the shipped size (F = 32, 720 x 1280, fast TTA) with by-name random weights.

    python tools/vfi_bench.py [--pairs 3] [--dtype bf16|fp16] [--height 720 --width 1280]

Prints ms per interpolated frame and the per-kernel time split (HIP events around every launch, like bench.py).
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--graph", action="store_true", help="also time a hipGraph replay of one inference (launch-overhead check)")
    a = ap.parse_args()
    from streamingt2v_amd import ops
    from streamingt2v_amd.ema_vfi import EMAVFI, VFIConfig
    from streamingt2v_amd.params import init_by_name
    ops.set_element_dtype(torch.float16 if a.dtype == "fp16" else torch.bfloat16)
    torch.set_grad_enabled(False)
    m = EMAVFI(VFIConfig())
    t0 = time.time()
    m.load_state_dict(init_by_name(m.spec(), seed=3), device="cuda")
    print(f"weights: {m.spec().numel() / 1e6:.1f} M parameters, packed in {time.time() - t0:.1f} s")
    g = torch.Generator().manual_seed(0)
    f0, f1 = (torch.rand(a.height, a.width, 3, generator=g).cuda() for _ in range(2))
    m.inference(f0, f1)                                     # warm-up: geometry tables, allocator
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.time()
    for _ in range(a.pairs):
        out, u8 = m.inference(f0, f1, want_uint8=True)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.pairs
    print(f"EMA-VFI {a.height}x{a.width} F=32 fast-TTA {a.dtype}: {dt * 1e3:.1f} ms per interpolated frame "
          f"({1 / dt:.2f} frames/s), peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, finite={bool(torch.isfinite(out).all())}")


    if a.graph:
        import time as _t
        t0 = _t.time()
        for _ in range(a.pairs):
            m.inference(f0, f1, want_uint8=True)
        cpu_issue = (_t.time() - t0) / a.pairs                      # host time to ISSUE one inference (no sync)
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m.inference(f0, f1, want_uint8=True)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gout, gu8 = m.inference(f0, f1, want_uint8=True)
        g.replay(); torch.cuda.synchronize()
        t0 = _t.time()
        for _ in range(a.pairs):
            g.replay()
        torch.cuda.synchronize()
        dg = (_t.time() - t0) / a.pairs
        print(f"host issue time {cpu_issue * 1e3:.1f} ms per inference; hipGraph replay {dg * 1e3:.1f} ms per interpolated frame "
              f"({1 / dg:.2f} frames/s), identical={bool(torch.equal(gout, out))}")


if __name__ == "__main__":
    main()
