#!/usr/bin/env python
"""Per-workgroup phase times of the 8-wave GEMM kernel (timing build: -DMICO_GEMM_ABLATE=7 exports mico_debug_phase_times).
    MICO_HIP_LIB=tools/probes/bin/libmico_abl7.so python tools/probes/gemm_phases.py
Stamps (s_memtime-class counter): 0 kernel entry, 1 prologue DMA issued, 2 K loop done, 3 epilogue instructions done."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mico_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("PHASES_M", "82240"))
if os.environ.get("PHASES_VARIANT"):
    _lib.set_gemm_variant(int(os.environ["PHASES_VARIANT"]))
for name, K, N in (("qkv", 1408, 4224), ("fc1", 1408, 6144), ("fc2", 6144, 1408)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (0.02 * torch.randn(N, K, device=dev)).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev)
    for _ in range(3):
        ops.gemm(x, w, y, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(x, w, y, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    ntiles = ((M + 255) // 256) * ((N + 255) // 256)
    n = min(ntiles, 8192)
    buf = (C.c_ulonglong * (n * 8))()
    fn = _lib.lib().mico_debug_phase_times
    fn.argtypes = [C.c_void_p, C.c_int]
    assert fn(buf, n * 8) == 0
    t8 = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
    t = t8[:, :4]
    bad = int((t == 0).any(axis=1).sum())
    if bad:
        print(f"   {bad} workgroups without stamps; first rows: {t[:3].tolist()}")
        t = t[~(t == 0).any(axis=1)]
    d = np.diff(t, axis=1)
    start = t[:, 0] - t[:, 0].min()
    span = (t[:, 3].max() - t[:, 0].min())
    print(f"{name}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us, {ntiles} tiles; counter span {span:.0f} ticks -> {e0.elapsed_time(e1) * 1e3 / span * 1e3:.2f} ns/tick")
    ns = e0.elapsed_time(e1) * 1e6 / span
    print(f"   per workgroup (us): prologue {d[:, 0].mean() * ns / 1e3:.2f}   K loop {d[:, 1].mean() * ns / 1e3:.2f}   epilogue {d[:, 2].mean() * ns / 1e3:.2f}"
          f"   total {(t[:, 3] - t[:, 0]).mean() * ns / 1e3:.2f};  rounds {ntiles / 256:.1f} x total = {(t[:, 3] - t[:, 0]).mean() * ns / 1e3 * ntiles / 256:.0f} us")
    print(f"   epilogue detail (us): barrier {(t8[:, 4] - t8[:, 2]).mean() * ns / 1e3:.2f}   first 64-row block {(t8[:, 5] - t8[:, 4]).mean() * ns / 1e3:.2f}   second {(t8[:, 3] - t8[:, 5]).mean() * ns / 1e3:.2f}   store drain after the last instruction {(t8[:, 6] - t8[:, 3]).mean() * ns / 1e3:.2f}")
    # gap between a workgroup's end and the start of the next workgroup that begins after it (dispatch + drain), from sorted starts
    order = np.sort(t[:, 0])
    ends = np.sort(t[:, 3])
    print(f"   first 256 starts spread {(order[255] - order[0]) * ns / 1e3:.2f} us; median start-to-start of successive rounds {(np.median(order[256:512]) - np.median(order[:256])) * ns / 1e3:.2f} us")
