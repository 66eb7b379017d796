"""Phase profile of the GEMM K loop (probe build with -DSVD_GEMM_PHASE_PROFILE, loaded through SVD_LIB_FILE): shader-clock cycles that wave 0 of
workgroup 0 spends in [wait vmcnt | barrier | K walk | compute (ds_read + MFMA + DMA issue) | epilogue], per K tile and per output tile.
    SVD_LIB_FILE=libsvdhip_pv_phase.so python tools/gemm_phase_profile.py 8,21"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "8,21").split(",")]
DT = torch.float16
# res: 0 none, 1 16-bit residual -> 16-bit output, 2 the fp32 residual stream (fp32 residual -> fp32 output, the stream kernel)
shapes = [("ff1 geglu L0", 460800, 2560, 320, 1, 0, None), ("ff1 geglu L1", 115200, 5120, 640, 1, 0, None), ("to_out L0 +R16", 460800, 320, 320, 0, 1, None),
          ("to_out L0 +R32", 460800, 320, 320, 0, 2, None), ("ff2 L0 +R16", 460800, 320, 1280, 0, 1, None), ("ff2 L0 +R32", 460800, 320, 1280, 0, 2, None),
          ("ff2 L2 +R16", 28800, 1280, 5120, 0, 1, None), ("conv 320 L0 +R16", 460800, 320, 2880, 0, 1, (320, 72, 128, 50))]
if len(sys.argv) > 2:
    shapes = [sh for sh in shapes if any(k in sh[0] for k in sys.argv[2].split(","))]
for (name, M, N, K, geglu, res, conv) in shapes:
    if conv:
        cin, H, W, Fr = conv
        a = torch.randn(Fr * H * W, cin, device="cuda").to(DT); M = Fr * H * W
    else:
        a = torch.randn(M, K, device="cuda").to(DT)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(DT)
    bias = torch.randn(N, device="cuda"); nout = N // 2 if geglu else N
    out = torch.empty(M, nout, device="cuda", dtype=torch.float32 if res == 2 else DT)
    R = torch.randn(M, nout, device="cuda").to(torch.float32 if res == 2 else DT) if res else None
    for cfg in cfgs:
        g = L.GemmArgs(); g.A, g.lda = a.data_ptr(), a.stride(0); g.W, g.ldw = w.data_ptr(), K; g.M, g.N, g.K = M, N, K; g.bias = bias.data_ptr()
        if res: g.R, g.ldr = R.data_ptr(), nout
        if res == 2: g.res_f32, g.out_mode = 1, L.OUT_F32
        if conv:
            g.a_mode = L.A_CONV3X3; g.cin, g.hin, g.win, g.hout, g.wout, g.stride, g.ups = cin, H, W, H, W, 1, 0
        g.zeros = ops.zeros_page(a.device).data_ptr(); g.C, g.ldc = out.data_ptr(), nout; g.epi_flags = geglu; g.tile_cfg = cfg; g.dtype = L.DTYPE_F16
        if L.lib.svd_gemm_config_valid(C.byref(g), cfg) != 1:
            continue
        dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
        g.dbg_cycles = dbg.data_ptr()
        for _ in range(3):
            assert L.lib.svd_gemm(C.byref(g), st) == 0
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize()
        c = dbg.tolist()
        bk = 64
        tiles = max(c[5], 1); nkt = tiles * (K // bk)
        kloop = c[0] + c[1] + c[2] + c[3]
        other = c[6] - kloop - c[4]                      # next tile's prologue requests + tile set-up + kernel start / end
        print(f"{name:16s} cfg{cfg:2d} {s.elapsed_time(e):.3f} ms | wave 0 of WG 0: {tiles} tiles, {c[6] / tiles:8.0f} cyc per tile at {c[6] / s.elapsed_time(e) / 1e6:.2f} GHz | per K tile (BK 64 units): "
              f"wait {c[0] / nkt:6.0f}  barrier {c[1] / nkt:6.0f}  walk {c[2] / nkt:4.0f}  compute {c[3] / nkt:6.0f} | per tile: K loop {kloop / tiles:8.0f}  epilogue {c[4] / tiles:8.0f}  "
              f"rest {other / tiles:7.0f} | next tile's prologue {c[7] / tiles:6.0f}  barrier before the epilogue {c[8] / tiles:6.0f}  epilogue phase A {c[9] / tiles:6.0f}  phase B {c[10] / tiles:6.0f}", flush=True)
