#!/usr/bin/env python
"""Micro-benchmark of mico_gemm at the ViT-g/14 shapes of BASELINE config 3 (M = 82240 token rows), all three orientations.
    python tools/gemm_bench.py [--iters 20] [--only fwd|dx|dw] [--m 82240]
Prints TFLOP/s per shape from HIP events on the launch stream (random bf16 data, not zero-filled)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mico_amd import ops  # noqa: E402
from mico_amd.functional import _split_k  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--m", type=int, default=82240)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--split-k", type=int, default=0, help="K split of the weight-gradient launches (0 = the library's choice)")
    ap.add_argument("--resid", action="store_true", help="forward GEMMs with the towers' output-projection epilogue: fp32 out = resid + row_scale * "
                                                          "(acc + bias), scattered through a frame map (kept-frame compaction)")
    ap.add_argument("--mx8", action="store_true", help="forward GEMMs on the block-scaled fp8 MFMA (+ the activation quantisation pass)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = torch.device("cuda:0")
    M = a.m
    shapes = [("qkv", 1408, 4224), ("proj", 1408, 1408), ("fc1", 1408, 6144), ("fc2", 6144, 1408)]
    res = []
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev).to(dt)
        w = (0.02 * torch.randn(N, K, device=dev)).to(dt)
        dy = torch.randn(M, N, device=dev).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        dx = torch.empty(M, K, device=dev, dtype=dt)
        dw = torch.zeros(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        sk = a.split_k
        if a.mx8:
            qx, qw = ops.quant_mx8(x), ops.quant_mx8(w)
        if a.resid:
            frames = M // 257
            stream = torch.randn(frames * 257 + 257 * 8, N, device=dev)
            fmap = torch.arange(frames, device=dev, dtype=torch.int32) + (torch.arange(frames, device=dev, dtype=torch.int32) // 8)   # skips every 9th frame
            fmap = fmap.clamp_max(stream.shape[0] // 257 - 1).contiguous()
            rscale = torch.full((stream.shape[0] // 257,), 1.25, device=dev)
            Mr = frames * 257
        cases = {
            "fwd": (lambda: ops.gemm_mx8(qx, qw, y, dtype=dt, bias=bias)) if a.mx8 else
                   ((lambda: ops.gemm(x[:Mr], w, stream, bias=bias, resid=stream, row_scale=rscale, rows_per_scale=257, row_map=fmap, rows_per_map=257))
                    if a.resid else (lambda: ops.gemm(x, w, y, bias=bias))),
            **({"quant": lambda: ops.quant_mx8(x)} if a.mx8 else {}),
            "dx": lambda: ops.gemm(dy, w, dx, tb=True, M=M, N=K, K=N),
            "dw": lambda: ops.gemm(dy, x, dw, ta=True, tb=True, M=N, N=K, K=M, accumulate=True, split_k=sk),
        }
        for cname, fn in cases.items():
            if (a.only and cname != a.only) or (a.mx8 and cname in ("dx", "dw")):
                continue
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            lib = ops._lib.lib()
            if hasattr(lib, "mico_debug_w4_prof"):   # timing build (-DMICO_GEMM_ABLATE=8)
                import ctypes
                buf = (ctypes.c_ulonglong * 4)()
                lib.mico_debug_w4_prof(buf)
                if buf[3]:
                    print(f"      per K-step of 32 (wave 0, s_memtime ticks): vmcnt wait {buf[0] / buf[3]:.0f}  barrier {buf[1] / buf[3]:.0f}  whole {buf[2] / buf[3]:.0f}")
            ms = e0.elapsed_time(e1) / a.iters
            tf = 2.0 * M * N * K / ms / 1e9
            res.append((name, cname, ms, tf))
            print(f"{name:5s} {cname:3s} M={M} N={N} K={K} split_k={sk if cname == 'dw' else 1}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
    tot_f = sum(2.0 * M * (4224 + 1408 + 6144 + 6144) * 1408 for _ in range(1))
    for c in ("fwd", "dx", "dw"):
        ms = sum(r[2] for r in res if r[1] == c)
        if ms:
            print(f"layer {c}: {ms:.3f} ms  -> {tot_f / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
