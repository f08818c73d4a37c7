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
    ap.add_argument("--variants", default="", help="comma list of mico_gemm_set_variant values to A/B in this process, interleaved per shape (e.g. 0,5)")
    ap.add_argument("--square", type=int, default=0, help="also time an n x n x n forward problem (the guide's 4096^3 / 8192^3 reference shapes)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = torch.device("cuda:0")
    M = a.m
    shapes = [("qkv", 1408, 4224), ("proj", 1408, 1408), ("fc1", 1408, 6144), ("fc2", 6144, 1408)]
    res = []
    variants = [int(v) for v in a.variants.split(",")] if a.variants else [None]
    if a.square:
        n = a.square
        x = torch.randn(n, n, device=dev).to(dt)
        w = torch.randn(n, n, device=dev).to(dt)
        y = torch.empty(n, n, device=dev, dtype=dt)
        for v in variants * 2:
            if v is not None:
                ops._lib.set_gemm_variant(v)
            for _ in range(3):
                ops.gemm(x, w, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                ops.gemm(x, w, y)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print(f"square fwd n={n} variant={v}: {ms:8.3f} ms  {2.0 * n ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
        del x, w, y
    ref_out = {}
    for name, K, N in shapes:
        ref_out.clear()
        x = torch.randn(M, K, device=dev).to(dt)
        w = (0.02 * torch.randn(N, K, device=dev)).to(dt)
        dy = torch.randn(M, N, device=dev).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        dx = torch.empty(M, K, device=dev, dtype=dt)
        dw = torch.zeros(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        sk = a.split_k
        auxt, t1 = ops.aux_buffer(M, N, K, dt, dev)
        auxt2, t2 = ops.aux_buffer(M, K, N, dt, dev)
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
            "dxaux": lambda: ops.gemm(dy, w, dx, tb=True, M=M, N=K, K=N, aux_in=x, act=ops.ACT_MUL_AUX),   # (fc2's dX times the stored GELU': x stands in for it)
            # the MLP pair, row-major and tiled aux (mico_gemm_epilogue::aux_tiled): forward with the GELU pair, dX times the stored GELU'
            "pair": lambda: ops.gemm(x, w, y, bias=bias, aux_out=dy, act=ops.ACT_GELU_SAVE_DERIV),
            "pairt": lambda: ops.gemm(x, w, y, bias=bias, aux_out=auxt, act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=True),
            "dxauxt": lambda: ops.gemm(dy, w, dx, tb=True, M=M, N=K, K=N, aux_in=auxt2, act=ops.ACT_MUL_AUX, aux_tiled=True),
            "dw": lambda: ops.gemm(dy, x, dw, ta=True, tb=True, M=N, N=K, K=M, accumulate=True, split_k=sk),
        }
        for cname, fn, var in [(c, f, v) for c, f in cases.items() for v in (variants * 2 if len(variants) > 1 else variants)]:
            if (a.only and cname not in a.only.split(",")) or (a.mx8 and cname in ("dx", "dxaux", "dw", "pair", "pairt", "dxauxt")) or (cname in ("dxaux", "pair", "pairt", "dxauxt") and cname not in a.only.split(",")) or (cname == "pairt" and not t1) or (cname == "dxauxt" and not t2):
                continue
            if var is not None:
                ops._lib.set_gemm_variant(var)
                cname = f"{cname}@{var}"
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
            chk = ""
            if var is not None and not a.resid and cname.split("@")[0] in ("fwd", "dx"):   # variants must agree (same k order: expect 0)
                out = (y if cname.startswith("fwd") else dx).float()
                key = (name, cname.split("@")[0])
                if key not in ref_out:
                    ref_out[key] = out.clone()
                else:
                    chk = f"  max|diff vs variant {variants[0]}| = {(out - ref_out[key]).abs().max().item():.3g}"
            print(f"{name:5s} {cname:3s} M={M} N={N} K={K} split_k={sk if cname == 'dw' else 1}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s{chk}", flush=True)
    tot_f = sum(2.0 * M * (4224 + 1408 + 6144 + 6144) * 1408 for _ in range(1))
    for c in sorted(set(r[1] for r in res)):
        rows = [r for r in res if r[1] == c]
        reps = max(1, len(rows) // len(shapes))
        ms = sum(r[2] for r in rows) / reps
        if ms and len(rows) % len(shapes) == 0:
            print(f"layer {c}: {ms:.3f} ms  -> {tot_f / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
