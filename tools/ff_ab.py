"""Time svd_ff_geglu_fused at the job's sizes (A/B between library builds: SVD_LIB_FILE=...).   python tools/ff_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
from streamingt2v_amd.video_model import pack_ff_fused
torch.manual_seed(0)
C, Hd = 320, 1280
img = pack_ff_fused(torch.randn(2 * Hd, C) * C ** -0.5, torch.randn(2 * Hd) * 0.1, torch.randn(C, Hd) * Hd ** -0.5).cuda()
b2 = torch.randn(C, device="cuda") * 0.1
for M in (460800, 129024, 1094400):
    x = torch.randn(M, C, device="cuda").to(ops.ELEM)
    r = torch.randn(M, C, device="cuda")
    s_ = torch.randn(M, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    te = torch.randn((M + 9215) // 9216, C, device="cuda")
    y0 = torch.randn(M, C, device="cuda")
    for _ in range(3):
        ops.layernorm(y0, g, b, addvec=te, rows_per_vec=9216)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.layernorm(y0, g, b, addvec=te, rows_per_vec=9216)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M:8d} {'svd_layernorm fp32 -> 16 bit (+vec)':34s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
    for name, kw in (("fp32 residual -> fp32", dict(residual=r, out_f32=True)), ("fp32 residual -> fp32 + LayerNorm", dict(residual=r, out_f32=True, ln=(g, b), ln_addvec=te, ln_rows_per_vec=9216)),
                     ("fp32 residual + blend -> 16 bit", dict(residual=r, blend=(0.3, s_), out_f32=False))):
        for _ in range(3):
            ops.ff_geglu_fused(x, img, Hd, b2, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.ff_geglu_fused(x, img, Hd, b2, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"M={M:8d} {name:34s} {ms * 1e3:8.1f} us  {6.0 * M * C * Hd / ms / 1e9:7.1f} TFLOP/s", flush=True)
