"""In-process A/B of a gemm epi_flags debug bit over representative shapes: python tools/gemm_ab_flag.py <bit>"""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
bit = 1 << int(sys.argv[1])
def t(fn, reps=9):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts)//2]
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
M=460800
for (N,K,ge) in [(2560,320,1),(320,320,0),(320,1280,0),(960,320,0),(640,320,0),(1280,1280,0),(5120,640,1)]:
    Mx = M if K<=320 or N==320 or N==1280 else M//4
    a=torch.randn(Mx,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")*K**-0.5).to(torch.bfloat16)
    bias=torch.randn(N,device="cuda"); out=torch.empty(Mx,N//2 if ge else N,device="cuda",dtype=torch.bfloat16)
    line=f"M={Mx} N={N} K={K} geglu={ge}:"
    for cfg in (1,2,5,8,17,18,19,20):
        r=[]
        for flag in (0, bit):
            g=L.GemmArgs(); g.A,g.lda=a.data_ptr(),K; g.W,g.ldw=w.data_ptr(),K; g.M,g.N,g.K=Mx,N,K; g.bias=bias.data_ptr()
            g.zeros=ops.zeros_page(a.device).data_ptr(); g.C,g.ldc=out.data_ptr(),out.shape[1]; g.epi_flags=flag|(1 if ge else 0); g.tile_cfg=cfg
            r.append(t(lambda: L.lib.svd_gemm(C.byref(g), st)))
        line+=f"  cfg{cfg}: {r[0]:.3f} / flag {r[1]:.3f}"
    print(line, flush=True)
