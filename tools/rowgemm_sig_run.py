"""Launch svd_rowgemm320 (to_out form: fp32 residual in, fp32 Y + 16-bit LayerNorm(Y) out, per-frame vector) a few times, alone, at M rows so that a rocprofv3
--pmc pass attributes FETCH_SIZE / WRITE_SIZE to this kernel (tools/pmc_round6.sh).      python tools/rowgemm_sig_run.py [M] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
from streamingt2v_amd.video_model import pack_rowgemm320
M = int(sys.argv[1]) if len(sys.argv) > 1 else 460800
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
w = torch.randn(320, 320) * 320 ** -0.5
img = pack_rowgemm320(w).cuda()
x = torch.randn(M, 320, device="cuda").to(ops.ELEM)
r = torch.randn(M, 320, device="cuda")
rv = torch.randn(M // 9216, 320, device="cuda")
g, b, bias = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda"), torch.randn(320, device="cuda") * 0.1
for _ in range(reps):
    y, yn = ops.rowgemm320(x, img, bias=bias, rowvec=rv, rows_per_vec=9216, residual=r, ln=(g, b))
torch.cuda.synchronize()
print("launched rowgemm320", M, reps, "x")
