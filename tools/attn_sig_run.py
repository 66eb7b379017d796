"""Launch the level-0 spatial self-attention of the job (50 frames x 9216 tokens x 5 heads, fp16) a few times, alone, for a rocprofv3 --pmc pass
(tools/pmc_round4.sh):  python tools/attn_sig_run.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames, n, heads = 50, 9216, 5
C = heads * 64
g = torch.Generator(device="cuda"); g.manual_seed(0)
q = torch.randn(frames * n, C, generator=g, device="cuda").to(torch.float16); k = torch.randn(frames * n, C, generator=g, device="cuda").to(torch.float16)
vt = torch.randn(frames, C, (n + 63) // 64 * 64, generator=g, device="cuda").to(torch.float16); o = torch.empty_like(q)
for _ in range(reps):
    ops.attn_spatial(q, k, vt, o, frames, n, heads)
torch.cuda.synchronize()
print("launched attn_spatial", frames, n, heads, reps, "x")
