"""Launch the level-0 fused feed-forward of the job (M = 460 800 rows = 50 frames @ 72x128, C = 320, hidden 1280, fp32 residual in / fp32 out, fp16) a few
times, alone, for a rocprofv3 --pmc pass (tools/pmc_ff.sh):  python tools/ff_sig_run.py [reps]      (SVD_FF_WAVES=8 selects the eight-wave form)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
from streamingt2v_amd.video_model import pack_ff_fused
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M, C, Hd = 460800, 320, 1280
g = torch.Generator(device="cuda"); g.manual_seed(0)
x = torch.randn(M, C, generator=g, device="cuda").to(torch.float16)
r = torch.randn(M, C, generator=g, device="cuda")
gc = torch.Generator(); gc.manual_seed(1)
img = pack_ff_fused(torch.randn(2 * Hd, C, generator=gc) * C ** -0.5, torch.randn(2 * Hd, generator=gc) * 0.2, torch.randn(C, Hd, generator=gc) * Hd ** -0.5).cuda()
b2 = torch.randn(C, generator=g, device="cuda") * 0.2
out = torch.empty_like(r)
for _ in range(reps):
    ops.ff_geglu_fused(x, img, Hd, b2, residual=r, out_f32=True, out=out)
torch.cuda.synchronize()
print("launched ff_geglu_fused", M, C, Hd, reps, "x")
