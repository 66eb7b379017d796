"""Same-process timing of explicit tile configurations on the job's GEMM shapes, with a bit checksum per output (every tile must produce the same bits:
the K order does not depend on the tile), for probe builds that carry experimental tiles (SVD_LIB_FILE):
    SVD_LIB_FILE=libsvdhip_pv_xpf.so python tools/gemm_cfg_ab.py 20,8,17,18,19,25,26,27 [reps]"""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
cfgs = [int(c) for c in sys.argv[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ops.set_element_dtype(torch.float16)
g = torch.Generator(device="cuda"); g.manual_seed(0)
# (name, M, N, K, geglu, residual, conv (cin, H, W, frames) | None)
shapes = [("ff1 GEGLU L0", 460800, 2560, 320, 1, 0, None), ("ff1 GEGLU L1", 115200, 5120, 640, 1, 0, None), ("ff1 GEGLU L2", 28800, 10240, 1280, 1, 0, None),
          ("qkv L0", 460800, 960, 320, 0, 0, None), ("to_out L0 +R", 460800, 320, 320, 0, 1, None), ("ff2 L0 +R", 460800, 320, 1280, 0, 1, None),
          ("ff2 L1 +R", 115200, 640, 2560, 0, 1, None), ("ff2 L2 +R", 28800, 1280, 5120, 0, 1, None), ("conv 640 L1 +R", 115200, 640, 5760, 0, 1, (640, 36, 64, 50))]
print("# lib:", os.environ.get("SVD_LIB_FILE", "libsvdhip.so"), " cfgs:", cfgs)
for name, M, N, K, geglu, res, conv in shapes:
    if conv:
        from streamingt2v_amd.video_model import pack_conv3x3
        cin, H, W, Fr = conv
        a = torch.randn(Fr * H * W, cin, generator=g, device="cuda").half()
        w = pack_conv3x3((torch.randn(N, cin, 3, 3, generator=g, device="cuda") * K ** -0.5).cpu()).half().cuda()
        kw = dict(conv=dict(cin=cin, frames=Fr, hin=H, win=W, hout=H, wout=W, stride=1, ups=0))
    else:
        a = torch.randn(M, K, generator=g, device="cuda").half()
        w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).half()
        kw = {}
    bias = torch.randn(N, generator=g, device="cuda")
    r = torch.randn(M, N, generator=g, device="cuda").half() if res else None
    row, sums = [], {}
    for cfg in cfgs:
        try:
            for _ in range(2):
                out = ops.gemm(a, w, bias=bias, geglu=bool(geglu), residual=r, tile_cfg=cfg, **kw)
        except Exception as e:      # tile cannot run this launch
            row.append(f"{cfg}: -"); continue
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for i in range(reps):
            out = ops.gemm(a, w, bias=bias, geglu=bool(geglu), residual=r, tile_cfg=cfg, **kw)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        sums[cfg] = (out.view(torch.int16).long() * torch.arange(1, out.numel() + 1, device="cuda").view_as(out).remainder(8191)).sum().item()
        row.append(f"{cfg}: {ts[reps // 2]:.4f}")
    same = len(set(sums.values())) == 1
    print(f"{name:15s} M={M} N={N} K={K} | " + "  ".join(row) + f" | ms (median of {reps}); bits identical across tiles: {same}" + ("" if same else f" {sums}"), flush=True)
