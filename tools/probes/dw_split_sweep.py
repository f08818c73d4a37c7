"""Weight-gradient GEMM (dW = dy^T x, producer/consumer kernel, split-K slabs) against a uniform K split (--split-k semantics of ops.gemm;
0 = the library's choice).  Each configuration is warmed up: the first timing in a process otherwise reads ~10 % low (clock ramp).
DW_TOKENS=<rows> changes the reduction length (the kept-token count varies per block in situ)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from mico_amd import ops
dev = torch.device("cuda:0")
M = int(os.environ.get("DW_TOKENS", "82240"))
def run(kin, nout, sk, iters=20):
    x = torch.randn(M, kin, device=dev).half(); dy = torch.randn(M, nout, device=dev).half()
    dw = torch.zeros(nout, kin, device=dev)
    fn = lambda: ops.gemm(dy, x, dw, ta=True, tb=True, M=nout, N=kin, K=M, accumulate=True, split_k=sk)
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"out={nout} in={kin} split={sk}: {ms:.3f} ms {2.0*M*kin*nout/ms/1e9:.1f} TF", flush=True)
for kin, nout in [(1408, 4224), (1280, 4224), (1408, 6144), (1280, 6144), (1408, 1408), (6144, 1408)]:
    for sk in [0, 2, 3, 4, 5, 6, 7, 8, 10]:
        run(kin, nout, sk)
