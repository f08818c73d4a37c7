#!/usr/bin/env python
"""Round 5: the 8-phase kernel's fast epilogue with 16-bit staging (MICO_P8_EPI16=1, the default) against the fp32-staged one (probe build with
-DMICO_P8_EPI16=0): outputs must agree BIT FOR BIT (same fp32 arithmetic per element, rounded once).  Run once per library:
    MICO_HIP_LIB=... python tools/probes/epi16_check.py save /tmp/a.pt ;  python tools/probes/epi16_check.py save /tmp/b.pt ;  ... compare /tmp/a.pt /tmp/b.pt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run():
    from mico_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for dt in (torch.float16, torch.bfloat16):
        for (M, N, K) in ((257 * 131 - 100, 1368, 1408), (65792, 4224, 1408), (40000, 6144, 1408), (33000, 1408, 6144)):
            g = torch.Generator(device="cpu").manual_seed(M + N + K)
            A = (0.5 * torch.randn(M, K, generator=g)).to(dev).to(dt)
            W = (0.05 * torch.randn(N, K, generator=g)).to(dev).to(dt)
            bias = torch.randn(N, generator=g).to(dev)
            tag = f"{dt}-{M}x{N}x{K}"
            y = torch.empty(M, N, device=dev, dtype=dt)
            ops.gemm(A, W, y, bias=bias, alpha=0.75)
            out[tag + "-lean"] = y.cpu()
            ops.gemm(A, W, y)
            out[tag + "-plain"] = y.cpu()
            ops.gemm(A, W, y, bias=bias, act=ops.ACT_GELU)
            out[tag + "-gelu"] = y.cpu()
            aux = torch.empty(M, N, device=dev, dtype=dt)
            ops.gemm(A, W, y, bias=bias, aux_out=aux, act=ops.ACT_GELU_SAVE_DERIV)
            out[tag + "-pair"] = y.cpu()
            out[tag + "-pair-aux"] = aux.cpu()
            Wt = W.t().contiguous()
            ops.gemm(A, Wt, y, tb=True, M=M, N=N, K=K)
            out[tag + "-dx"] = y.cpu()
            ops.gemm(A, Wt, y, tb=True, M=M, N=N, K=K, aux_in=aux, act=ops.ACT_MUL_AUX, alpha=0.5)      # (aux: the pair launch's GELU')
            out[tag + "-dxaux"] = y.cpu()
            if K % 128 == 0:      # the MX-fp8 kernels (same epilogues; the persistent form gemm_p8pmx_kernel)
                qa, qw = ops.quant_mx8(A), ops.quant_mx8(W)
                ops.gemm_mx8(qa, qw, y, dtype=dt, bias=bias)
                out[tag + "-mx8-lean"] = y.cpu()
                ops.gemm_mx8(qa, qw, y, dtype=dt, bias=bias, aux_out=aux, act=ops.ACT_GELU_SAVE_DERIV)
                out[tag + "-mx8-pair"] = y.cpu()
                out[tag + "-mx8-pair-aux"] = aux.cpu()
    return out


if __name__ == "__main__":
    if sys.argv[1] == "save":
        torch.save(run(), sys.argv[2])
    else:
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = 0
        for k in a:
            same = torch.equal(a[k].view(torch.int16), b[k].view(torch.int16))
            if not same:
                bad += 1
                d = (a[k].float() - b[k].float()).abs()
                print("DIFF", k, "max", d.max().item(), "count", int((d > 0).sum()))
        print(f"{len(a)} outputs compared, {bad} differ")
        sys.exit(1 if bad else 0)
