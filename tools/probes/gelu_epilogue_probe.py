import torch, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mico_amd import ops, _lib
dev=torch.device('cuda:0'); dt=torch.float16
M,N,K=184000,6144,1408
A=(0.5*torch.randn(M,K,device=dev)).to(dt); W=(0.05*torch.randn(N,K,device=dev)).to(dt); bias=torch.randn(N,device=dev)
out=torch.empty(M,N,device=dev,dtype=dt); aux=torch.empty(M,N,device=dev,dtype=dt)
def t(fn,n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
lib=_lib.lib()
for name,fn in (("pair",lambda: ops.gemm(A,W,out,bias=bias,aux_out=aux,act=ops.ACT_GELU_SAVE_DERIV)),("gelu",lambda: ops.gemm(A,W,out,bias=bias,act=ops.ACT_GELU)),("lean",lambda: ops.gemm(A,W,out,bias=bias))):
    ms=t(fn); k=getattr(lib,'mico_debug_last_gemm_kernel',None)
    print(name, f"{ms:.3f} ms", f"{2*M*N*K/ms/1e9:.0f} TFLOP/s", k() if k else '')
