"""Where does a GEMM's time go?  Re-times one shape with parts of the kernel switched off (probe bits in epi_flags):
   bit 30 = no LDS-DMA loads, bit 29 = no ds_read/MFMA, bit 27 = no epilogue.  If full ~ loads_only + mfma_only + epi_only
   nothing overlaps; if full ~ max(...) everything does."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = 460800
NOL, NOM, NOE = 1 << 30, 1 << 29, 1 << 27
VAR = [("full", 0), ("no_epi", NOE), ("loads_only", NOM | NOE), ("mfma_only", NOL | NOE), ("epi_only", NOL | NOM), ("loads+epi", NOM), ("mfma+epi", NOL)]
if os.environ.get("PROBE_FULL_ONLY"):
    VAR = [("full", 0)]
if os.environ.get("PROBE_STAGGER"):
    VAR = [("full", 0)] + [(f"stag{s}", s << 20) for s in (4, 8, 16, 24, 32, 48, 63)]
shapes = [(2560, 320, 1, 0), (320, 320, 0, 1), (1280, 1280, 0, 0), (320, 2880, 0, 0), (1280, 5120, 0, 1), (512, 4608, 0, 0)]
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 1, 2, 23]
for (N, K, geglu, res) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda"); nout = N // 2 if geglu else N
    out = torch.empty(M, nout, device="cuda", dtype=torch.bfloat16); R = torch.randn(M, nout, device="cuda").to(torch.bfloat16) if res else None
    flops = 2.0 * M * N * K
    for cfg in cfgs:
        line = []
        for name, bits in VAR:
            g = L.GemmArgs(); g.A, g.lda = a.data_ptr(), K; g.W, g.ldw = w.data_ptr(), K; g.M, g.N, g.K = M, N, K; g.bias = bias.data_ptr()
            if res: g.R, g.ldr = R.data_ptr(), nout
            g.zeros = ops.zeros_page(a.device).data_ptr(); g.C, g.ldc = out.data_ptr(), nout; g.epi_flags = geglu | bits; g.tile_cfg = cfg
            best = 1e9
            for _ in range(6):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); rc = L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize()
                best = min(best, s.elapsed_time(e))
            line.append(f"{name} {best:.3f}")
        full = float(line[0].split()[1])
        print(f"N={N} K={K} geglu={geglu} res={res} cfg{cfg}: " + " | ".join(line) + f"  [{flops / full / 1e9:.0f} TF]", flush=True)
