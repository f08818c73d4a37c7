"""Execution settings and the 16-bit weight cache of the MI355X engine.

compute dtype: torch.float16 is the parity configuration (embeddings/logits within 1e-3 of the fp32 reference),
torch.bfloat16 the throughput configuration BASELINE.json's metric is quoted in.  Parameters stay fp32 (state-dict
compatible with the reference); a 16-bit copy of every GEMM weight is (re)materialised only when the parameter's
version counter changes, i.e. once per optimiser step.
"""
import contextlib

import torch

from . import ops


class _Cfg:
    compute_dtype = torch.bfloat16
    # internal 16-bit gradients are carried multiplied by grad_scale (fp16 only: guards against underflow); every
    # function de-scales what it hands back, so autograd sees true-scale fp32 gradients.
    grad_scale = {torch.float16: 4096.0, torch.bfloat16: 1.0}


CFG = _Cfg()


def compute_dtype():
    return CFG.compute_dtype


def grad_scale():
    return CFG.grad_scale[CFG.compute_dtype]


def set_compute_dtype(dtype):
    assert dtype in (torch.float16, torch.bfloat16)
    CFG.compute_dtype = dtype


@contextlib.contextmanager
def precision(dtype):
    old = CFG.compute_dtype
    set_compute_dtype(dtype)
    try:
        yield
    finally:
        CFG.compute_dtype = old


_W16 = {}


def w16(key, params, build):
    """16-bit derived weight, cached on (key, dtype) and invalidated by the source parameters' version counters."""
    dt = CFG.compute_dtype
    ver = tuple((p.data_ptr(), p._version) for p in params)
    hit = _W16.get((key, dt))
    if hit is not None and hit[0] == ver:
        return hit[1]
    with torch.no_grad():
        t = build(dt)
    _W16[(key, dt)] = (ver, t)
    return t


def clear_weight_cache():
    _W16.clear()


def cast_weight(w, dt, k_pad=None, n_pad=None):
    """fp32 [N, K...] parameter -> 16-bit [N(_pad), K_pad] (zero padded), K flattened."""
    w2 = w.detach().reshape(w.shape[0], -1)
    N, K = w2.shape
    kp = k_pad or ops.pad8(K)
    out = torch.zeros((n_pad or N, kp), dtype=dt, device=w.device) if (n_pad and n_pad != N) else torch.empty(
        (N, kp), dtype=dt, device=w.device)
    ops.cast_f32_to_16(w2.contiguous(), out[:N], cols=K, cols_pad=kp)
    return out
