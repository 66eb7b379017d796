"""Does the leading dimension of A / W matter (L2-channel / HBM-bank aliasing of power-of-two row strides)?  Times plain GEMM shapes of the job with
dense rows (ld = K) and with rows padded by `pad` elements.    python tools/gemm_stride_probe.py [cfg list] [pads]"""
import sys, os, ctypes as C, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "8,21,18").split(",")]
pads = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "0,64,32,8").split(",")]
DT = torch.float16
shapes = [("ff2 L2", 28800, 1280, 5120, 0), ("ff2 L1", 115200, 640, 2560, 0), ("ff2 L0", 460800, 320, 1280, 0), ("proj L2", 28800, 1280, 1280, 0),
          ("ff1 geglu L2", 28800, 10240, 1280, 1), ("ff1 geglu L0", 460800, 2560, 320, 1), ("qkv-like L1", 115200, 640, 640, 0)]
for (name, M, N, K, geglu) in shapes:
    nout = N // 2 if geglu else N
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, nout, device="cuda", dtype=DT)
    R = None if geglu else torch.randn(M, nout, device="cuda").to(DT)
    args = {}
    keep = []
    for pad in pads:
        a = torch.randn(M, K + pad, device="cuda").to(DT)
        w = (torch.randn(N, K + pad, device="cuda") * K ** -0.5).to(DT)
        keep += [a, w]
        for cfg in cfgs:
            g = L.GemmArgs(); g.A, g.lda = a.data_ptr(), K + pad; g.W, g.ldw = w.data_ptr(), K + pad; g.M, g.N, g.K = M, N, K; g.bias = bias.data_ptr()
            if R is not None: g.R, g.ldr = R.data_ptr(), nout
            g.zeros = ops.zeros_page(a.device).data_ptr(); g.C, g.ldc = out.data_ptr(), nout; g.epi_flags = geglu; g.tile_cfg = cfg; g.dtype = L.DTYPE_F16
            if L.lib.svd_gemm_config_valid(C.byref(g), cfg) == 1:
                args[(pad, cfg)] = g
    times = {k: [] for k in args}
    for r in range(6):
        for k, g in args.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rc = L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize()
            assert rc == 0, (k, rc)
            if r: times[k].append(s.elapsed_time(e))
    flops = 2.0 * M * N * K
    for cfg in cfgs:
        line = "  ".join(f"pad{pad:3d}: {statistics.median(times[(pad, cfg)]):.3f} ms ({flops / statistics.median(times[(pad, cfg)]) / 1e9:5.0f} TF)" for pad in pads if (pad, cfg) in times)
        print(f"{name:13s} M={M} N={N} K={K} cfg{cfg:2d}: {line}", flush=True)
