"""Per-phase cycle profile of one wave of the persistent GEMM (s_memtime counters): wait / barrier / issue / compute / epilogue."""
import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamingt2v_amd import ops, lib as L
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
M=460800
for (N,K,flags,res,cfgs) in [(2560,320,1,0,(8,1,2,19,20)),(320,320,0,1,(8,1,2,19,20)),(1280,1280,0,0,(8,1,2,19,20)),(320,2880,0,0,(8,2,19,20)),(1280,5120,0,1,(8,1,2,19,20))]:
    a=torch.randn(M,K,device="cuda").to(torch.bfloat16); w=(torch.randn(N,K,device="cuda")*K**-0.5).to(torch.bfloat16)
    bias=torch.randn(N,device="cuda"); nout=N//2 if flags&1 else N
    out=torch.empty(M,nout,device="cuda",dtype=torch.bfloat16); R=torch.randn(M,nout,device="cuda").to(torch.bfloat16) if res else None
    for cfg in cfgs:
        dbg=torch.zeros(8,dtype=torch.int64,device="cuda")
        g=L.GemmArgs(); g.A,g.lda=a.data_ptr(),K; g.W,g.ldw=w.data_ptr(),K; g.M,g.N,g.K=M,N,K; g.bias=bias.data_ptr()
        if res: g.R,g.ldr=R.data_ptr(),nout
        g.zeros=ops.zeros_page(a.device).data_ptr(); g.C,g.ldc=out.data_ptr(),nout; g.epi_flags=flags; g.tile_cfg=cfg; g.dbg_cycles=dbg.data_ptr()
        for _ in range(2): L.lib.svd_gemm(C.byref(g), st)
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); L.lib.svd_gemm(C.byref(g), st); e.record(); e.synchronize()
        d=dbg.tolist(); tiles=max(d[5],1); tot=sum(d[:5])
        print(f"N={N} K={K} geglu={flags&1} res={res} cfg{cfg}: {s.elapsed_time(e):.3f} ms | per tile cycles: wait {d[0]//tiles} barrier {d[1]//tiles} issue {d[2]//tiles} compute {d[3]//tiles} epilogue {d[4]//tiles} (tiles {tiles}, total/tile {tot//tiles})", flush=True)
