import sys, os, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best=1e9
    for _ in range(reps):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); best=min(best,s.elapsed_time(e))
    return best
M=460800
for (N,K) in [(2560,320),(320,320),(320,1280),(1280,1280)]:
    a=torch.randn(M,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")*K**-0.5).to(torch.bfloat16)
    out=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
    for cfg in (1,5,8):
        res=[]
        for flags in (0, 1<<30, 1<<29, (1<<29)|(1<<30)):
            args=L.GemmArgs(); args.A,args.lda=a.data_ptr(),K; args.W,args.ldw=w.data_ptr(),K; args.M,args.N,args.K=M,N,K
            args.zeros=ops.zeros_page(a.device).data_ptr(); args.C,args.ldc=out.data_ptr(),N; args.epi_flags=flags; args.tile_cfg=cfg
            st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
            res.append(t(lambda: L.lib.svd_gemm(C.byref(args), st)))
        print(f"N={N} K={K} cfg{cfg}: full {res[0]:.3f}  no-store {res[1]:.3f}  1-ktile {res[2]:.3f}  1-ktile+no-store {res[3]:.3f}", flush=True)
x=torch.empty(M*2560,device="cuda",dtype=torch.bfloat16)
print("fill 2.36GB: %.3f ms"%t(lambda: x.zero_()))
y=torch.empty_like(x); print("copy 2.36GB: %.3f ms"%t(lambda: y.copy_(x)))
