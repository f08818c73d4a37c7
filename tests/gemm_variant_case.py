"""Body of tests/test_kernels_gpu.py::test_gemm_chip_filling_kernels, run as a script in a process of its own against the PROBE build of the
library (MICO_HIP_LIB = tools/probes/bin/libmico_variants.so, `make -C mico_amd/csrc variants`): the product library has no kernel-routing
switch, so forcing each large-tile kernel onto one problem needs the build that has it.
    python tests/gemm_variant_case.py <variant> <f16|bf16>"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rel_err(a, b):      # (as in tests/test_kernels_gpu.py)
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12)).item()


def tol(dtype, k=1.0):
    return (2e-3 if dtype == torch.float16 else 1.6e-2) * k


def run(variant, dtype):
    cuda = torch.device("cuda:0")

    from mico_amd import ops, _lib
    torch.manual_seed(5)
    frames, rows_per = 131, 257
    M, N, K = frames * rows_per - 100, 1368, 1408
    A = (0.5 * torch.randn(M, K, device=cuda)).to(dtype)
    W = (0.05 * torch.randn(N, K, device=cuda)).to(dtype)
    bias = torch.randn(N, device=cuda)
    acc = A.float() @ W.float().t()
    old = _lib.set_gemm_variant(variant)
    try:
        y = torch.empty(M, N, device=cuda, dtype=dtype)
        ops.gemm(A, W, y, bias=bias)
        assert _lib.lib().mico_gemm_last_kernel() == {12: 1, 5: 6, 8: 7, 10: 8}[variant]
        assert rel_err(y, acc + bias) < tol(dtype)
        # dX orientation: the weight read reduction-major
        Wt = W.t().contiguous()
        y2 = torch.empty(M, N, device=cuda, dtype=dtype)
        ops.gemm(A, Wt, y2, tb=True, M=M, N=N, K=K)
        assert rel_err(y2, acc) < tol(dtype)
        # residual scatter: the compact rows of the kept frames go to frames fmap[f] of the fp32 stream, scaled per frame, in place
        nf = (M + rows_per - 1) // rows_per
        fmap = (torch.arange(nf, device=cuda, dtype=torch.int32) * 3 // 2).contiguous()          # skips every third frame
        stream = torch.randn((int(fmap[-1]) + 1) * rows_per, N, device=cuda)
        rs = torch.rand(int(fmap[-1]) + 1, device=cuda) + 0.5
        ref = stream.clone()
        rows = (fmap.long().repeat_interleave(rows_per) * rows_per + torch.arange(rows_per, device=cuda).repeat(nf))[:M]
        ref[rows] += (acc + bias) * rs[fmap.long()].repeat_interleave(rows_per)[:M, None]
        ops.gemm(A, W, stream, bias=bias, resid=stream, row_scale=rs, rows_per_scale=rows_per, row_map=fmap, rows_per_map=rows_per)
        assert rel_err(stream, ref) < 1e-5 * math.sqrt(K)
        # the MLP pair
        gd = torch.empty(M, N, device=cuda, dtype=dtype)
        a2 = torch.empty(M, N, device=cuda, dtype=dtype)
        ops.gemm(A, W, a2, bias=bias, aux_out=gd, act=ops.ACT_GELU_SAVE_DERIV)
        pre = acc + bias
        gp32 = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
        assert rel_err(a2, F.gelu(pre)) < tol(dtype) and rel_err(gd, gp32) < tol(dtype)
        dh = torch.empty(M, N, device=cuda, dtype=dtype)
        ops.gemm(A, Wt, dh, tb=True, M=M, N=N, K=K, aux_in=gd, act=ops.ACT_MUL_AUX, alpha=0.5)
        assert rel_err(dh, 0.5 * acc * gd.float()) < tol(dtype)
        # two k-segments (x W_hi + x W_lo of the split-weights precision mode): A read twice, B = [hi | lo]
        if dtype == torch.float16:
            Wf = 0.05 * torch.randn(N, K, device=cuda)
            hi = Wf.to(dtype)
            lo = (Wf - hi.float()).to(dtype)
            Wcat = torch.cat((hi, lo), dim=1).contiguous()
            y3 = torch.empty(M, N, device=cuda, dtype=torch.float32)
            ops.gemm(A, Wcat, y3, ksegs=(K, [0, 0], [0, K]))
            assert rel_err(y3, A.float() @ Wf.t()) < 2e-5 * math.sqrt(K)
    finally:
        _lib.set_gemm_variant(old)




def run_persistent(dtype):
    """The persistent form of the 8-phase kernel (default routing for 16-bit outputs with more than 256 tiles) against the one-tile kernel
    (variant 16 = persistent form off): bit-identical outputs - lean with / without bias and alpha, GELU, the GELU pair (both outputs), the dX
    orientation, the wrapped-A two-segment product of the head-split blocks; ragged M, N with a half-width edge tile column, N % 128 != 0 (the
    dX launch then stays on the one-tile kernel by routing)."""
    from mico_amd import ops, _lib
    cuda = torch.device("cuda:0")
    for (M, N, K) in ((257 * 131 - 100, 1408, 1408), (70001, 4224, 1408), (40000, 1368, 2816)):
        g = torch.Generator().manual_seed(M + N)
        A = (0.5 * torch.randn(M, K, generator=g)).to(cuda).to(dtype)
        W = (0.05 * torch.randn(N, K, generator=g)).to(cuda).to(dtype)
        bias = torch.randn(N, generator=g).to(cuda)
        Wt = W.t().contiguous()
        outs = {}
        for variant in (0, 16):
            _lib.set_gemm_variant(variant)
            res = []
            y = torch.empty(M, N, device=cuda, dtype=dtype)
            ops.gemm(A, W, y, bias=bias, alpha=0.5)
            assert _lib.lib().mico_gemm_last_kernel() == 8
            res.append(y.clone())
            ops.gemm(A, W, y)
            res.append(y.clone())
            ops.gemm(A, W, y, bias=bias, act=ops.ACT_GELU)
            res.append(y.clone())
            aux = torch.empty(M, N, device=cuda, dtype=dtype)
            ops.gemm(A, W, y, bias=bias, aux_out=aux, act=ops.ACT_GELU_SAVE_DERIV)
            res += [y.clone(), aux.clone()]
            ops.gemm(A, Wt, y, tb=True, M=M, N=N, K=K)
            res.append(y.clone())
            if dtype == torch.float16 and K % 128 == 0:      # x W_hi + x W_lo as one product over [hi | lo] (a_wrap)
                k2 = K // 2
                ops.gemm(A[:, :k2].contiguous(), W, y, ksegs=(k2, [0, 0], [0, k2]))
                res.append(y.clone())
            outs[variant] = res
        _lib.set_gemm_variant(0)
        for i, (a, b) in enumerate(zip(outs[0], outs[16])):
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (M, N, K, i, (a.float() - b.float()).abs().max().item())


if __name__ == "__main__":
    dt = torch.float16 if sys.argv[2] == "f16" else torch.bfloat16
    if sys.argv[1] == "persistent":
        run_persistent(dt)
    else:
        run(int(sys.argv[1]), dt)
    print("OK")
