import os, sys, torch, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mico_amd import ops, _lib
cuda = torch.device("cuda:0")
def run(ta, tb, M, N, K, dtype, reps=4, lda=None, ldb=None):
    g = torch.Generator(device="cuda").manual_seed(5)
    ldm, ldn, ldk = ops.pad8(M), ops.pad8(N), ops.pad8(K)
    A = torch.randn((K, lda or ldm) if ta else (M, lda or ldk), device=cuda, generator=g).to(dtype)
    B = torch.randn((K, ldb or ldn) if tb else (N, ldb or ldk), device=cuda, generator=g).to(dtype)
    print("strides", A.stride(0), B.stride(0), end="  ")
    Af = (A[:, :M].t() if ta else A[:, :K]).float(); Bf = (B[:, :N].t() if tb else B[:, :K]).float()
    ref = Af @ Bf.t()
    for r in range(reps):
        out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float32)
        ops.gemm(A, B, out, ta=ta, tb=tb, M=M, N=N, K=K, dtype=dtype)
        d = (out - ref).abs()
        bad = (d > 1e-3 * ref.abs().max()) | out.isnan()
        nb = int(bad.sum())
        msg = ""
        if nb:
            idx = bad.nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            msg = f" bad rows {rows.min().item()}..{rows.max().item()} (uniq {rows.unique().numel()}) cols {cols.min().item()}..{cols.max().item()} (uniq {cols.unique().numel()})  first {idx[0].tolist()} got {out[idx[0][0], idx[0][1]].item():.3f} want {ref[idx[0][0], idx[0][1]].item():.3f}"
        print(f"ta={int(ta)} tb={int(tb)} {M}x{N}x{K} {dtype} rep{r}: max err {d.max().item() / ref.abs().max().item():.2e} bad {nb}{msg}", flush=True)
import sys
vs = [int(x) for x in sys.argv[1:]] or [3]
for v in vs:
    _lib.set_gemm_variant(v)
    print("variant", v)
    for dt in (torch.bfloat16,):
        run(False, True, 8232, 2048, 1408, dt, reps=4)
def run_sk(dtype):
    torch.manual_seed(6)
    rows, n_out, n_in = 257 * 37, 1408, 2816
    dy = (0.1 * torch.randn(rows, n_out, device=cuda)).to(dtype)
    x = torch.randn(rows, n_in, device=cuda).to(dtype)
    for sk in (0, 1, 3):
        dw = torch.randn(n_out, n_in, device=cuda)
        ref = dw + 0.25 * (dy.float().t() @ x.float())
        ops.gemm(dy, x, dw, ta=True, tb=True, M=n_out, N=n_in, K=rows, accumulate=True, alpha=0.25, split_k=sk)
        d = (dw - ref).abs()
        print(f"  split-K TT {dtype} split_k={sk}: rel err {(d.max() / ref.abs().max()).item():.2e} (tol {2e-5 * rows ** 0.5:.2e}) bad {(d > 1e-2 * ref.abs().max()).sum().item()}")
for v in vs:
    _lib.set_gemm_variant(v)
    print("variant", v); run_sk(torch.bfloat16)
