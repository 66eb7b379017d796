"""Launch ONE plain-GEMM signature (as printed in bench.py's roofline.traffic.signature) a few times, alone, so that a rocprofv3 --pmc
pass attributes FETCH_SIZE / WRITE_SIZE to exactly this signature (tools/pmc_signature.sh).
    python tools/gemm_sig_run.py m0_M460800_N2560_K320_s0_u0_e1_o0 [reps] [dtype]"""
import os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
sig = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dt = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float16
m = re.match(r"m(\d)_M(\d+)_N(\d+)_K(\d+)_s(\d)_u(\d)_e(\d+)_o(\d)", sig)
mode, M, N, K, _, _, epi, omode = (int(v) for v in m.groups())
assert mode == 0, "plain GEMM signatures only"
ops.set_element_dtype(dt)
g = torch.Generator(device="cuda"); g.manual_seed(0)
a = torch.randn(M, K, generator=g, device="cuda").to(dt)
w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(dt)
bias = torch.randn(N, generator=g, device="cuda")
geglu = bool(epi & 1)
# the layer-0 feed-forward of the job: GEGLU proj has no residual; every other plain GEMM of that size adds one
res = None if geglu else torch.randn(M, N, generator=g, device="cuda").to(dt)
for _ in range(reps):
    out = ops.gemm(a, w, bias=bias, geglu=geglu, silu=bool(epi & 2), residual=res, out_f32=(omode == 1))
torch.cuda.synchronize()
print("launched", sig, reps, "x; out", tuple(out.shape), out.dtype)
