"""Times the spatial self-attention kernel on the two level-0 shapes (SVD 50 x 9216 x 5 heads; enhancer 76 x 14400 x 5)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
for frames, n, heads in ((50, 9216, 5), (76, 14400, 5), (50, 2304, 10)):
    C = heads * 64
    q = torch.randn(frames * n, C, device="cuda").to(torch.bfloat16); k = torch.randn_like(q)
    vt = torch.randn(frames, C, (n + 63) // 64 * 64, device="cuda").to(torch.bfloat16); o = torch.empty_like(q)
    for _ in range(2): ops.attn_spatial(q, k, vt, o, frames, n, heads)
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.attn_spatial(q, k, vt, o, frames, n, heads); e.record(); e.synchronize(); best = min(best, s.elapsed_time(e))
    print(f"attn_spatial {frames}x{n}x{heads}: {best:.3f} ms  {4.0 * frames * heads * n * n * 64 / best / 1e9:.0f} TF")
