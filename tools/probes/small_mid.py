import torch, sys, os
sys.path.insert(0, "/root/repo")
from mico_amd import ops, _lib
dev = torch.device("cuda:0")
dt = torch.float16
def run(M, N, K, variant, tb=False, iters=20):
    _lib.set_gemm_variant(variant)
    x = torch.randn(M, K, device=dev).to(dt)
    w = (0.02 * torch.randn((K, N) if tb else (N, K), device=dev)).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    f = (lambda: ops.gemm(x, w, y, tb=True, M=M, N=N, K=K)) if tb else (lambda: ops.gemm(x, w, y))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K} tb={tb} variant={variant} kernel={_lib.lib().mico_gemm_last_kernel()}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s")
for M in (256, 1024, 65536, 65792):
    for v in (12, 15, 13):
        run(M, 6144, 1408, v)
run(65792, 1408, 6144, 12); run(65792, 1408, 6144, 13); run(65792, 1408, 6144, 14)
