"""Time the three GEGLU feed-forward GEMMs of the stage-1 forward (levels 0-2: M = 50 frames x tokens, N = 8 C, K = C), alone, with HIP events:
the A/B of the epilogue's GELU form (round 4: gelu_erf_f2 on v_pk_fma_f32 against the per-element gelu_erf_f; SVD_LIB_FILE selects the build).
    python tools/geglu_ab.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ops.set_element_dtype(torch.float16)
g = torch.Generator(device="cuda"); g.manual_seed(0)
print("# lib:", os.environ.get("SVD_LIB_FILE", "libsvdhip.so"))
for M, C in ((460800, 320), (115200, 640), (28800, 1280)):
    a = torch.randn(M, C, generator=g, device="cuda").half()
    w = (torch.randn(8 * C, C, generator=g, device="cuda") * C ** -0.5).half()
    bias = torch.randn(8 * C, generator=g, device="cuda")
    for _ in range(3):
        out = ops.gemm(a, w, bias=bias, geglu=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        out = ops.gemm(a, w, bias=bias, geglu=True)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    fl = 2.0 * M * 8 * C * C
    chk = (out.view(torch.int16).long() * torch.arange(1, out.numel() + 1, device="cuda").view_as(out).remainder(8191)).sum().item()
    med = ts[reps // 2]
    print(f"GEGLU M={M} N={8 * C} K={C}: median {med:.4f} ms  min {ts[0]:.4f} ms  ({fl / med / 1e9:.0f} TFLOP/s at the median)  bit checksum {chk}")
# the short-K residual GEMMs of the same blocks (16-bit residual -> 16-bit output): to_out / proj_out and ff2
for M, N, K in ((460800, 320, 320), (460800, 320, 1280), (115200, 640, 640), (460800, 640, 320)):
    a = torch.randn(M, K, generator=g, device="cuda").half()
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn(M, N, generator=g, device="cuda").half()
    for _ in range(3):
        out = ops.gemm(a, w, bias=bias, residual=res)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        out = ops.gemm(a, w, bias=bias, residual=res)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    chk = (out.view(torch.int16).long() * torch.arange(1, out.numel() + 1, device="cuda").view_as(out).remainder(8191)).sum().item()
    print(f"linear+R M={M} N={N} K={K}: median {ts[reps // 2]:.4f} ms  min {ts[0]:.4f} ms  ({2.0 * M * N * K / ts[reps // 2] / 1e9:.0f} TFLOP/s at the median)  bit checksum {chk}")
