#!/usr/bin/env python
"""Micro-benchmark of the fused attention kernels at the BASELINE config-3 shapes (ViT-g: 320 frames x 16 heads x 257 tokens x
hd 88, fused QKV buffer; BERT cross: 192 x 12 heads, 77 queries x 1285 keys, hd 64)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mico_amd import ops  # noqa: E402


def run(name, B, H, Sq, Sk, hd, fused, iters=10):
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    D = H * hd
    if fused:
        qkv = torch.randn(B, Sq, 3 * D, device=dev).to(dt)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        st = dict(q_strides=(Sq * 3 * D, 3 * D), k_strides=(Sk * 3 * D, 3 * D), v_strides=(Sk * 3 * D, 3 * D))
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        q = torch.randn(B, Sq, D, device=dev).to(dt)
        k = torch.randn(B, Sk, D, device=dev).to(dt)
        v = torch.randn(B, Sk, D, device=dev).to(dt)
        st = dict(q_strides=(Sq * D, D), k_strides=(Sk * D, D), v_strides=(Sk * D, D))
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    o = torch.empty(B, Sq, D, device=dev, dtype=dt)
    do = torch.randn(B, Sq, D, device=dev).to(dt)
    lse = torch.empty(B, H, Sq, device=dev)
    delta = torch.empty(B, H, Sq, device=dev)
    kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=hd ** -0.5, o_strides=(Sq * D, D), **st)
    flops = 4.0 * B * H * Sq * Sk * hd
    for label, fn, mult in (("fwd", lambda: ops.attn_fwd(q, k, v, o, lse, **kw), 1.0),
                            ("bwd", lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, delta, **kw), 2.5)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:10s} {label}: {ms:8.3f} ms  {flops * mult / ms / 1e9:7.1f} TFLOP/s (algorithmic)", flush=True)


if __name__ == "__main__":
    run("vit_g", 320, 16, 257, 257, 88, True)
    run("vit_g_256", 320, 16, 256, 256, 88, True)     # what the 257th token (ragged fifth tile in both directions) costs
    run("vit_g_272", 320, 16, 272, 272, 88, True)
    run("vit_l14", 128, 16, 257, 257, 64, True)       # the EVA02 towers (hd 64): L/14 ...
    run("vit_b16", 256, 12, 197, 197, 64, True)       # ... and B/16 (197 tokens: ragged / dead key blocks in the one-pass backward)
    run("bert_cross", 192, 12, 77, 1285, 64, False)
    run("bert_self", 192, 12, 77, 77, 64, True)
