"""A/B of GEMM tile configurations on the job's dominant signatures (plain + implicit-GEMM conv), interleaved in ONE process
(guide methodology rule 24): for every shape, every config is timed `reps` times round-robin; median and min reported.
   python tools/gemm_probe2.py 8,21,22,25 [reps]
variants: full | no_epi (bit 27) | mfma_only (bits 30+27: no LDS-DMA, no epilogue) -- see gemm_impl.inc probe bits.  The probe bits exist only in builds made
with -DSVD_GEMM_PROBES (make -C streamingt2v_amd/csrc gvariant NAME=probe PROBE_DEFS=-DSVD_GEMM_PROBES PV_CFGS=..., then SVD_LIB_FILE=libsvdhip_pv_probe.so);
in the product library only the `full` column means anything -- and the switches themselves cost ~10 % of the K loop, so A/B decisions use probe-free
variant builds (tools/probe_kloop.sh)."""
import sys, os, ctypes as C, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
from streamingt2v_amd.video_model import pack_conv3x3
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
NOL, NOM, NOE = 1 << 30, 1 << 29, 1 << 27
GEN = 1 << 26          # general K-loop path only (no lean iterations)
VAR = [("full", 0), ("full_gen", GEN), ("no_epi", NOE), ("mfma_only", NOL | NOE), ("mfma_only_gen", NOL | NOE | GEN)]
cfgs = [int(c) for c in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
DT = torch.float16
# (name, M, N, K, geglu, residual, conv)   conv = (cin, H, W, frames)
shapes = [("ff1 geglu L0", 460800, 2560, 320, 1, 0, None), ("proj L0", 460800, 320, 320, 0, 1, None), ("ff2 L0", 460800, 320, 1280, 0, 1, None),
          ("ff1 geglu L1", 115200, 5120, 640, 1, 0, None), ("proj L2", 28800, 1280, 1280, 0, 1, None), ("ff2 L2", 28800, 1280, 5120, 0, 1, None),
          ("conv 320 L0", 460800, 320, 2880, 0, 1, (320, 72, 128, 50)), ("conv 640 L1", 115200, 640, 5760, 0, 1, (640, 36, 64, 50)),
          ("conv 1280 L2", 28800, 1280, 11520, 0, 1, (1280, 18, 32, 50)), ("vae conv 128", 8 * 576 * 1024 // 4, 128, 1152, 0, 0, (128, 288, 512, 8))]
for (name, M, N, K, geglu, res, conv) in shapes:
    if conv:
        cin, H, W, Fr = conv
        a = torch.randn(Fr * H * W, cin, device="cuda").to(DT); M = Fr * H * W
    else:
        a = torch.randn(M, K, device="cuda").to(DT)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(DT)
    bias = torch.randn(N, device="cuda"); nout = N // 2 if geglu else N
    out = torch.empty(M, nout, device="cuda", dtype=DT); R = torch.randn(M, nout, device="cuda").to(DT) if res else None
    flops = 2.0 * M * N * K
    args = {}
    for cfg in cfgs:
        for vn, bits in VAR:
            g = L.GemmArgs(); g.A, g.lda = a.data_ptr(), a.stride(0); g.W, g.ldw = w.data_ptr(), K; g.M, g.N, g.K = M, N, K; g.bias = bias.data_ptr()
            if res: g.R, g.ldr = R.data_ptr(), nout
            if conv:
                g.a_mode = L.A_CONV3X3; g.cin, g.hin, g.win, g.hout, g.wout, g.stride, g.ups = cin, H, W, H, W, 1, 0
            g.zeros = ops.zeros_page(a.device).data_ptr(); g.C, g.ldc = out.data_ptr(), nout; g.epi_flags = geglu | bits; g.tile_cfg = cfg
            g.dtype = L.DTYPE_F16
            if L.lib.svd_gemm_config_valid(C.byref(g), cfg) != 1:
                continue
            args[(cfg, vn)] = g
    times = {k: [] for k in args}
    for r in range(reps + 1):
        for k, g in args.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); rc = L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize()
            assert rc == 0, (k, rc)
            if r: times[k].append(s.elapsed_time(e))
    for cfg in cfgs:
        if (cfg, "full") not in times: continue
        line = " | ".join(f"{vn} med {statistics.median(times[(cfg, vn)]):.3f} min {min(times[(cfg, vn)]):.3f}" for vn, _ in VAR)
        print(f"{name:14s} M={M} N={N} K={K} cfg{cfg:2d}: {line}  [{flops / statistics.median(times[(cfg, 'full')]) / 1e9:.0f} TF]", flush=True)
