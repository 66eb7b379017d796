"""Times the per-pixel temporal attention on the job's shapes (CFG 2 x 25 frames; levels 0-3; CAM 25 x 7 keys).  Run twice for the A/B:
    python tools/attn_temporal_bench.py            # MFMA kernel (T <= 64)
    SVD_ATTN_TEMPORAL_VALU=1 python tools/attn_temporal_bench.py      # the VALU kernel it replaced"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops
ops.set_element_dtype(torch.float16)
tag = "VALU" if os.environ.get("SVD_ATTN_TEMPORAL_VALU") else "MFMA"
for (B, tq, tk, pix, heads) in ((2, 25, 25, 9216, 5), (2, 25, 25, 2304, 10), (2, 25, 25, 576, 20), (2, 25, 25, 144, 20), (2, 25, 7, 9216, 5),
                             (2, 38, 38, 14400, 5), (2, 38, 38, 3600, 10)):      # the enhancer's 38-frame windows @ 90x160 / 45x80
    C = heads * 64
    qkv = torch.randn(B * tq * pix, 3 * C, device="cuda").to(torch.float16)
    kv = torch.randn(B * tk * pix, 2 * C, device="cuda").to(torch.float16)
    o = torch.empty(B * tq * pix, C, device="cuda", dtype=torch.float16)
    q, k, v = (qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]) if tq == tk else (qkv[:, :C], kv[:, :C], kv[:, C:])
    for _ in range(2): ops.attn_temporal(q, k, v, o, B, tq, tk, pix, heads)
    best = 1e9
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); ops.attn_temporal(q, k, v, o, B, tq, tk, pix, heads); e.record(); e.synchronize(); best = min(best, s.elapsed_time(e))
    nbytes = 2.0 * B * pix * C * (2 * tq + 2 * tk)
    print(f"[{tag}] attn_temporal B{B} {tq}x{tk} pix {pix} heads {heads}: {best * 1e3:7.1f} us  {nbytes / best / 1e9:6.2f} TB/s (q,k,v read + o written once)", flush=True)
