import sys, torch
sys.path.insert(0, "/root/repo")
from mico_amd import ops
dev = torch.device("cuda:0")
M = 65792
def run(kin, nout, cs, iters=20):
    x = torch.randn(M, kin, device=dev).half(); dy = torch.randn(M, nout, device=dev).half()
    dw = torch.zeros(nout, kin, device=dev); b = torch.zeros(nout, device=dev)
    fn = lambda: ops.gemm(dy, x, dw, ta=True, tb=True, M=nout, N=kin, K=M, accumulate=True, split_k=0, colsum_out=b if cs else None)
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"out={nout} in={kin} colsum={cs}: {ms:.3f} ms {2.0*M*kin*nout/ms/1e9:.1f} TF", flush=True)
for rep in range(2):
    for kin, nout in [(1408, 4224), (1408, 1408), (1408, 6144), (6144, 1408)]:
        for cs in (False, True):
            run(kin, nout, cs)
