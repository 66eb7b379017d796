"""Start de-phasing span of the persistent GEMM (epi_flags bits 20..25, units of 2048 cycles; 0 = the launcher's choice: 32 for 8-wave 256x256-class tiles with
many short-K tiles per workgroup) on the short-K shapes of the job.    python tools/gemm_dephase_probe.py 20,21"""
import sys, os, ctypes as C, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "20,21").split(",")]
spans = [0, 1, 8, 16, 24, 32, 48, 63]
DT = torch.float16
shapes = [("ff1 geglu L0", 460800, 2560, 320, 1, 0), ("ff1 geglu L1", 115200, 5120, 640, 1, 0), ("qkv L0", 460800, 640, 320, 0, 0), ("proj L0", 460800, 320, 320, 0, 1),
          ("ff2 L0", 460800, 320, 1280, 0, 1), ("ff2 L1", 115200, 640, 2560, 0, 1)]
for (name, M, N, K, geglu, res) in shapes:
    a = torch.randn(M, K, device="cuda").to(DT); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(DT)
    bias = torch.randn(N, device="cuda"); nout = N // 2 if geglu else N
    out = torch.empty(M, nout, device="cuda", dtype=DT); R = torch.randn(M, nout, device="cuda").to(DT) if res else None
    args = {}
    for cfg in cfgs:
        for sp in spans:
            g = L.GemmArgs(); g.A, g.lda = a.data_ptr(), K; g.W, g.ldw = w.data_ptr(), K; g.M, g.N, g.K = M, N, K; g.bias = bias.data_ptr()
            if res: g.R, g.ldr = R.data_ptr(), nout
            g.zeros = ops.zeros_page(a.device).data_ptr(); g.C, g.ldc = out.data_ptr(), nout; g.epi_flags = geglu | (sp << 20); g.tile_cfg = cfg; g.dtype = L.DTYPE_F16
            if L.lib.svd_gemm_config_valid(C.byref(g), cfg) == 1: args[(cfg, sp)] = g
    times = {k: [] for k in args}
    for r in range(6):
        for k, g in args.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rc = L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize(); assert rc == 0
            if r: times[k].append(s.elapsed_time(e))
    for cfg in cfgs:
        if (cfg, 0) in times:
            print(f"{name:13s} cfg{cfg}: " + "  ".join(f"span {sp:2d}: {statistics.median(times[(cfg, sp)]):.3f}" for sp in spans), flush=True)
