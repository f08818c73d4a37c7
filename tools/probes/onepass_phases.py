#!/usr/bin/env python
"""Per-wave cycle counts of the one-pass tower attention backward (attn_bwd_onepass_kernel) in a timing build:
    make -C mico_amd/csrc phases   ->  tools/probes/bin/libmico_attnph.so   (-DMICO_ATTN_PHASES)
    MICO_HIP_LIB=tools/probes/bin/libmico_attnph.so python tools/probes/onepass_phases.py
Slots per item (cycles, mean over the first 256 workgroups): 0 item start (K image, V rows, two barriers), 1 key 256 for chunk 0,
2 phase 1 (S, dP, P, dS, dV, dK), 3 the barrier, 4 commit chunk G+2 (waits for its loads), 5 request chunk G+3, 6 phase 2 (dQ tile /
key 256 for the next chunk), 7 item end (dK / dV stores)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mico_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
names = ["item start", "key256 c0", "phase 1", "barrier", "commit", "issue", "phase 2", "item end"]
for (B, H, S, hd) in ((320, 16, 257, 88), (320, 16, 256, 88)):
    D = H * hd
    dt = torch.float16
    qkv = torch.randn(B, S, 3 * D, device=dev).to(dt)
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    o = torch.empty(B, S, D, device=dev, dtype=dt)
    do = torch.randn(B, S, D, device=dev).to(dt)
    lse = torch.empty(B, H, S, device=dev)
    delta = torch.empty(B, H, S, device=dev)
    dqkv = torch.empty_like(qkv)
    kw = dict(B=B, H=H, Sq=S, Sk=S, hd=hd, scale=hd ** -0.5, q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D),
              v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
    ops.attn_fwd(q, k, v, o, lse, **kw)
    run = lambda: ops.attn_bwd(q, k, v, o, do, lse, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], delta, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    n = 4096
    buf = (C.c_ulonglong * (n * 8))()
    fn = _lib.lib().mico_debug_attn_phases
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, n * 8) == 0
    tw = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8).astype(np.float64)[:256]
    items = B * H / 256
    print(f"S={S}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us, {items:.0f} items per workgroup; cycles per item and wave:")
    print("          " + "  ".join(f"{nm:>10s}" for nm in names) + "       total")
    for w in range(8):
        print(f"   wave {w}: " + "  ".join(f"{tw[:, w, i].mean() / items:10.0f}" for i in range(8)) + f"  {tw[:, w, :].sum(1).mean() / items:10.0f}")
