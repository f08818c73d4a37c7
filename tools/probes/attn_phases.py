#!/usr/bin/env python
"""Per-wave cycle counts of the attention forward kernel's phases (timing build: -DMICO_ATTN_PHASES exports
mico_debug_attn_phases).   MICO_HIP_LIB=tools/probes/bin/libmico_attnph.so python tools/probes/attn_phases.py
Slots: 0 wait at the top barrier, 1 commit registers -> LDS, 2 second barrier, 3 issue next tile's global loads,
4 S^T = K Q^T (LDS fragment reads + MFMA), 5 softmax, 6 O^T += V^T P^T."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mico_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
names = ["barrier1", "commit", "barrier2", "fetch-issue", "S=KQ^T", "softmax", "PV", "-"]
res_names = ["top barrier", "merge", "commit(+wait)", "barrier 2", "prefetch part 0", "17th block", "key steps (+prefetch, next Q)", "stores"]
for (B, H, S, hd) in ((320, 16, 256, 88), (320, 16, 257, 88)):
    D = H * hd
    qkv = torch.randn(B, S, 3 * D, device=dev).bfloat16()
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    o = torch.empty(B, S, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=dev)
    kw = dict(B=B, H=H, Sq=S, Sk=S, hd=hd, scale=hd ** -0.5, q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D),
              v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
    for _ in range(3):
        ops.attn_fwd(q, k, v, o, lse, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attn_fwd(q, k, v, o, lse, **kw)
    e1.record()
    torch.cuda.synchronize()
    n = 4096
    buf = (C.c_ulonglong * (n * 8))()
    fn = _lib.lib().mico_debug_attn_phases
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, n * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
    t = t[t.sum(1) > 0]
    tot = t.sum(1)
    print(f"S={S}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us; {len(t)} workgroups sampled; cycles per workgroup (wave 0) mean {tot.mean():.0f} "
          f"(min {tot.min():.0f}, max {tot.max():.0f})")
    res = S <= 260 and not os.environ.get("MICO_ATTN_NORES")
    if res:   # the resident kernel records every wave of the first 512 workgroups: rows = (workgroup, wave)
        tw = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8).astype(np.float64)[:256]
        for w in range(8):
            print(f"   wave {w}: " + "  ".join(f"{tw[:, w, i].mean() / (B * H / 256):.0f}" for i in range(8)))   # K/V-resident persistent kernel: per workgroup totals over its items
    items = B * H / 256 if res else 1
    print("   " + "  ".join(f"{nm} {t[:, i].mean() / items:.0f}" for i, nm in enumerate(res_names if res else names)) + ("   (cycles per item)" if res else ""))
