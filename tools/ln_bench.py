#!/usr/bin/env python
"""Micro-benchmark of the LayerNorm kernels at the ViT-g tower shapes (320 frames x 257 tokens x 1408, fp32 residual stream):
achieved HBM bandwidth of the forward (-> 16-bit GEMM operand, plain and as the compacting DropPath gather with the stream
copy) and of the backward (fp32 dy, fp32 x, in-place residual-gradient update, dgamma / dbeta), next to a device copy."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mico_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    F, N, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 320), 257, 1408   # 717 = one rank of configs[3] after stochastic depth
    M = F * N
    dt = torch.bfloat16
    x = torch.randn(M, D, device=dev)
    g, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
    y16 = torch.empty(M, D, device=dev, dtype=dt)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, out16=y16, mean=mean, rstd=rstd, dtype=dt))
    print(f"ln fwd plain        : {ms * 1e3:7.1f} us  {M * D * 6 / ms / 1e6:7.1f} GB/s (6 B/elem)")
    fmap = torch.arange(F, device=dev, dtype=torch.int32)
    xc = torch.empty_like(x)
    ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, out16=y16, mean=mean, rstd=rstd, dtype=dt, frame_map=fmap, rows_per_frame=N, x_copy=xc))
    print(f"ln fwd gather+copy  : {ms * 1e3:7.1f} us  {M * D * 10 / ms / 1e6:7.1f} GB/s (10 B/elem)")
    dy = torch.randn(M, D, device=dev)
    gr = torch.randn(M, D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ms = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx_add=gr, dx32=gr, dgamma=dg, dbeta=db, dtype=dt))
    print(f"ln bwd (tower form) : {ms * 1e3:7.1f} us  {M * D * 16 / ms / 1e6:7.1f} GB/s (16 B/elem)")
    ms = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx_add=gr, dx32=gr, dgamma=dg, dbeta=db, dtype=dt, frame_map=fmap, rows_per_frame=N))
    print(f"ln bwd scatter      : {ms * 1e3:7.1f} us  {M * D * 16 / ms / 1e6:7.1f} GB/s (16 B/elem)")
    # the in-situ form of round 3: 16-bit dy, frame scatter, and the next branch's 16-bit operand written along (gradient hand-over)
    dy16 = dy.to(dt)
    g16n = torch.empty(M, D, device=dev, dtype=dt)
    fsc = torch.ones(F, device=dev)
    ms = timeit(lambda: ops.layernorm_bwd(dy16, x, g, mean, rstd, dx_add=gr, dx32=gr, dgamma=dg, dbeta=db, dtype=dt, frame_map=fmap, rows_per_frame=N,
                                          dx16=g16n, scale16=1.0, dx16_dst=fmap, dx16_frame_scale=fsc))
    print(f"ln bwd in situ      : {ms * 1e3:7.1f} us  {M * D * 16 / ms / 1e6:7.1f} GB/s (16 B/elem: dy 2, x 4, g 4 + 4, next operand 2)")
    # round 5 (activation diet level 3): the forward leaves fp16 normalised rows instead of the fp32 copy, the backward and the recomputing
    # forward read them
    xh = torch.empty(M, D, device=dev, dtype=torch.float16)
    ms = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, out16=y16, mean=mean, rstd=rstd, dtype=dt, frame_map=fmap, rows_per_frame=N, xhat16=xh))
    print(f"ln fwd gather+xhat16: {ms * 1e3:7.1f} us  {M * D * 8 / ms / 1e6:7.1f} GB/s (8 B/elem)")
    ms = timeit(lambda: ops.layernorm_fwd(xh, g, b, 1e-6, out16=y16, dtype=dt, x_normalized=True))
    print(f"ln fwd from xhat16  : {ms * 1e3:7.1f} us  {M * D * 4 / ms / 1e6:7.1f} GB/s (4 B/elem)")
    ms = timeit(lambda: ops.layernorm_bwd(dy16, xh, g, None, rstd, dx_add=gr, dx32=gr, dgamma=dg, dbeta=db, dtype=dt, frame_map=fmap, rows_per_frame=N,
                                          dx16=g16n, scale16=1.0, dx16_dst=fmap, dx16_frame_scale=fsc, x_normalized=True))
    print(f"ln bwd in situ xhat : {ms * 1e3:7.1f} us  {M * D * 14 / ms / 1e6:7.1f} GB/s (14 B/elem: dy 2, xhat 2, g 4 + 4, next operand 2)")
    ms = timeit(lambda: xc.copy_(x))
    print(f"device copy fp32    : {ms * 1e3:7.1f} us  {M * D * 8 / ms / 1e6:7.1f} GB/s (8 B/elem)")
    ms = timeit(lambda: y16.copy_(x))
    print(f"device cast to bf16 : {ms * 1e3:7.1f} us  {M * D * 6 / ms / 1e6:7.1f} GB/s (6 B/elem)")
    gb = torch.randn(M, D, device=dev).to(dt)
    out = torch.zeros(D, device=dev)
    ms = timeit(lambda: ops.colsum(gb, out, scale=1.0, accumulate=True))
    print(f"colsum bf16 [M,1408]: {ms * 1e3:7.1f} us  {M * D * 2 / ms / 1e6:7.1f} GB/s (2 B/elem)")
    gb = torch.randn(M, 6144, device=dev).to(dt)
    out = torch.zeros(6144, device=dev)
    ms = timeit(lambda: ops.colsum(gb, out, scale=1.0, accumulate=True))
    print(f"colsum bf16 [M,6144]: {ms * 1e3:7.1f} us  {M * 6144 * 2 / ms / 1e6:7.1f} GB/s (2 B/elem)")


if __name__ == "__main__":
    main()
