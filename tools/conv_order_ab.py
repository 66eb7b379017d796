"""In-process A/B of the implicit-GEMM K order (taps innermost vs taps outermost) on the model's conv shapes."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
def t(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); return ts[len(ts)//2]
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (Fr,H,W,cin,cout) in [(50,72,128,320,320),(50,36,64,640,640),(50,18,32,1280,1280),(50,72,128,640,320),(8,576,1024,128,128),(8,288,512,256,256)]:
    M=Fr*H*W; x=torch.randn(M,cin,device="cuda").to(torch.bfloat16); w=(torch.randn(cout,9*cin,device="cuda")*(9*cin)**-0.5).to(torch.bfloat16)
    out=torch.empty(M,cout,device="cuda",dtype=torch.bfloat16)
    line=f"conv {Fr}x{H}x{W} {cin}->{cout}:"
    for cfg in (2,8,17,18,20):
        r=[]
        for flag in (0, 1<<28):
            a=L.GemmArgs(); a.A,a.lda=x.data_ptr(),cin; a.W,a.ldw=w.data_ptr(),9*cin; a.M,a.N,a.K=M,cout,9*cin; a.a_mode=1; a.cin=cin
            a.hin,a.win,a.hout,a.wout=H,W,H,W; a.stride=1; a.zeros=ops.zeros_page(x.device).data_ptr(); a.C,a.ldc=out.data_ptr(),cout
            a.epi_flags=flag; a.tile_cfg=cfg
            r.append(t(lambda: L.lib.svd_gemm(C.byref(a), st)))
        line+=f"  cfg{cfg}: inner {r[0]:.3f} outer {r[1]:.3f} ({2.0*M*cout*9*cin/r[0]/1e9:.0f} vs {2.0*M*cout*9*cin/r[1]/1e9:.0f} TF)"
    print(line, flush=True)
