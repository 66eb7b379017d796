"""Go / no-go probe of svd_rowgemm320 (csrc/rowgemm.hip) on the MI355X: time and algorithmic bandwidth of the row-owning projection (+ fused LayerNorm) against the
launches it replaces (svd_gemm with fp32 residual in / out, then svd_layernorm of the fp32 result) at the job's sizes.

    python tools/rowgemm_probe.py            -> profiles/r06_rowgemm_probe.txt (copy from gpurun_out/)

Review's bar: M = 460 800, fp32 R in / fp32 Y + 16-bit LN(Y) out at >= 4.5 TB/s algorithmic."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops  # noqa: E402
from streamingt2v_amd.video_model import pack_rowgemm320  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def main():
    torch.manual_seed(0)
    C = 320
    w = torch.randn(C, C) * C ** -0.5
    img = pack_rowgemm320(w).cuda()
    wd = w.to(ops.ELEM).cuda()
    bias = torch.randn(C, device="cuda") * 0.1
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    print(f"{'case':62s} {'rowgemm320':>12s} {'gemm':>10s} {'layernorm':>10s} {'two/three launches':>18s}  alg. TB/s (rowgemm | replaced)")
    for M, pix in ((460800, 9216), (129024, 9216), (1094400, 14400)):
        x = torch.randn(M, C, device="cuda").to(ops.ELEM)
        r = torch.randn(M, C, device="cuda")
        rv = torch.randn(M // pix, C, device="cuda")
        for name, kw, ln in (("proj_in: no residual, + LN", dict(), True), ("to_out: + rowvec + fp32 residual, + LN", dict(rowvec=rv, rows_per_vec=pix, residual=r), True),
                             ("proj_out: + fp32 residual", dict(residual=r), False)):
            t_new = timeit(lambda: ops.rowgemm320(x, img, bias=bias, ln=(g, b) if ln else None, **kw))
            t_g = timeit(lambda: ops.gemm(x, wd, bias=bias, out_f32=True, **kw))
            y = ops.gemm(x, wd, bias=bias, out_f32=True, **kw)
            t_l = timeit(lambda: ops.layernorm(y, g, b)) if ln else 0.0
            by_new = M * C * (2 + 4 + (4 if "residual" in kw else 0) + (2 if ln else 0))
            by_old = M * C * (2 + 4 + (4 if "residual" in kw else 0)) + (M * C * 6 if ln else 0)
            print(f"M={M:8d} {name:50s} {t_new * 1e3:9.1f} us {t_g * 1e3:7.1f} us {t_l * 1e3:7.1f} us {(t_g + t_l) * 1e3:15.1f} us  "
                  f"{by_new / t_new / 1e9:5.2f} | {by_old / (t_g + t_l) / 1e9:5.2f}", flush=True)
        del x, r, y


if __name__ == "__main__":
    main()
