"""Times GroupNorm (stats + apply) and LayerNorm on level-0 / level-1 shapes; prints achieved HBM GB/s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
def t(fn, n=5):
    for _ in range(2): fn()
    best = 1e9
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); best = min(best, s.elapsed_time(e))
    return best
for frames, pix, C in ((50, 9216, 320), (50, 2304, 640), (50, 576, 1280), (50, 9216, 640)):
    x = torch.randn(frames * pix, C, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    nbytes = x.numel() * 2
    ms = t(lambda: ops.groupnorm(x, frames, pix, g, b, 1e-5, silu=True))
    print(f"groupnorm+silu {frames}x{pix}x{C}: {ms*1e3:.0f} us  ({3 * nbytes / ms / 1e6:.0f} GB/s for read x2 + write)")
    ms = t(lambda: ops.layernorm(x, g, b))
    print(f"layernorm      {frames}x{pix}x{C}: {ms*1e3:.0f} us  ({2 * nbytes / ms / 1e6:.0f} GB/s for read + write)")
