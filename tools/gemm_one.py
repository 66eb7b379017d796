import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
M, N, K, cfg = 460800, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, w, bias=bias, out=out, tile_cfg=cfg)
torch.cuda.synchronize()
