#!/usr/bin/env python
"""Calibration probe (NOT on the product path): what does the vendor library (hipBLASLt behind torch.matmul) reach on the ViT-g/14 layer
shapes this repo's hand-written GEMMs run, on the same box in the same process?  Prints TFLOP/s per shape and orientation for
torch.matmul and for mico_gemm, so that the remaining headroom of the hand-written kernels is a measured number, not a guess.
    python tools/probes/hipblaslt_ref.py [--m 184269] [--dtype fp16] [--iters 10]
Run under `rocprofv3 --kernel-trace --stats` to see which library kernels (tile shapes) were picked."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mico_amd import ops  # noqa: E402


def timeit(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=717 * 257)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = torch.device("cuda:0")
    M = a.m
    tot = {}
    for name, K, N in [("qkv", 1408, 4224), ("proj", 1408, 1408), ("fc1", 1408, 6144), ("fc2", 6144, 1408)]:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (0.02 * torch.randn(N, K, device=dev)).to(dt)
        dy = torch.randn(M, N, device=dev).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        dx = torch.empty(M, K, device=dev, dtype=dt)
        dw32 = torch.zeros(N, K, device=dev)
        dw16 = torch.empty(N, K, device=dev, dtype=dt)
        wt = w.t()
        dyt = dy.t()
        cases = {
            "fwd": (lambda: torch.matmul(x, wt, out=y), lambda: ops.gemm(x, w, y)),
            "dx": (lambda: torch.matmul(dy, w, out=dx), lambda: ops.gemm(dy, w, dx, tb=True, M=M, N=K, K=N)),
            "dw": (lambda: torch.matmul(dyt, x, out=dw16), lambda: ops.gemm(dy, x, dw32, ta=True, tb=True, M=N, N=K, K=M, accumulate=True, split_k=0)),   # (split_k = 0: the library's own K split, as functional.linear_wgrad calls it)
        }
        for c, (lib_fn, our_fn) in cases.items():
            for who, fn in (("hipblaslt", lib_fn), ("mico", our_fn), ("hipblaslt", lib_fn), ("mico", our_fn)):
                ms = timeit(fn, a.iters)
                tf = 2.0 * M * N * K / ms / 1e9
                tot.setdefault((c, who), []).append(ms)
                print(f"{name:5s} {c:3s} {who:9s} M={M} N={N} K={K}: {ms:8.3f} ms {tf:7.1f} TFLOP/s", flush=True)
        del x, w, dy, y, dx, dw32, dw16
    flop = 2.0 * M * (4224 + 1408 + 6144 + 6144) * 1408
    for (c, who), v in sorted(tot.items()):
        ms = sum(v) / 2
        print(f"layer {c:3s} {who:9s}: {ms:.3f} ms -> {flop / ms / 1e9:.1f} TFLOP/s")
    for n in (4096, 8192):
        p = torch.randn(n, n, device=dev).to(dt)
        q = torch.randn(n, n, device=dev).to(dt)
        r = torch.empty(n, n, device=dev, dtype=dt)
        ms_l = timeit(lambda: torch.matmul(p, q.t(), out=r), a.iters)
        ms_o = timeit(lambda: ops.gemm(p, q, r), a.iters)
        print(f"square {n}: hipblaslt {2.0 * n ** 3 / ms_l / 1e9:7.1f}  mico {2.0 * n ** 3 / ms_o / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
