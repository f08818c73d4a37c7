"""Raw kernel launchers: torch tensors in (device memory + stream plumbing only), libmico_hip.so C-ABI calls out.

No autograd here and no fallback: every function requires CUDA(HIP) tensors and raises MicoHipError otherwise.
All launches go to torch's current stream.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import F16, BF16, F32, ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_DERIV, ACT_MUL_AUX, GemmEpilogue, AttnParams, MicoHipError, check

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}


def dt_code(dtype):
    return _DT[dtype]


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise MicoHipError("mico_amd kernels need device tensors (got a CPU tensor); there is no CPU fallback")
    return t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-launch HIP-event timing of the GEMM kernel (bench.py's roofline leg).  Events are recorded on the
    stream the kernel is launched on (torch's current stream)."""

    def __init__(self, variants=((0, 0), (0, 1), (1, 1))):
        self.variants = set(variants)
        self.records = []   # ((ta, tb, kernel), flops, start_event, end_event); kernel: 0 small tile, 1 8-wave, 2 producer/consumer
        self.detail = []    # per record: (M, N, K, epilogue signature) - bench.py --gemm-detail

    def summary(self):
        out = {}
        for var, flops, e0, e1 in self.records:
            d = out.setdefault(var, dict(launches=0, flops=0.0, ms=0.0))
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += e0.elapsed_time(e1)
        return out


GEMM_TIMER = None


def pad8(n):
    return (n + 7) // 8 * 8


_SPLITK_WS = {}
SPLITK_WS_STREAMS = 4   # scratch buffers kept: the most recently used (device, stream) pairs
SPLITK_SLABS = True   # False: weight-gradient launches get no scratch and fall back to fp32 atomics (kept tested: the C-ABI makes the scratch optional)
SPLITK_WS_CAP = 320 << 20   # bytes: room for 8+ slabs of the towers' largest layer gradient (6144 x 1408 fp32 = 35 MB); a smaller scratch only
                            # bounds the split count (the library falls back to fewer splits / atomics cleanly)


def _splitk_scratch(nbytes, device):
    """One fp32 scratch buffer per (device, stream), grown on demand and kept.  The weight-gradient kernel writes its split-K partial
    tiles into it and the reduction pass reads them back: launches on ONE stream use it one after the other; launches on different
    streams (or from threads with different current streams) must not share it - hence the stream in the key."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _SPLITK_WS.pop(key, None)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    _SPLITK_WS[key] = t                          # (re-inserted last: the dict is the LRU order)
    while len(_SPLITK_WS) > SPLITK_WS_STREAMS:   # streams created per step would otherwise each pin up to SPLITK_WS_CAP bytes for ever
        _SPLITK_WS.pop(next(iter(_SPLITK_WS)))
    return t


def release_scratch():
    """Drop the cached split-K scratch buffers (runtime.reset())."""
    _SPLITK_WS.clear()


def gemm(A, B, out, *, ta=False, tb=False, M=None, N=None, K=None, bias=None, aux_out=None, aux_in=None, act=ACT_NONE,
         row_scale=None, rows_per_scale=0, resid=None, pos=None, pos_rows=0, remap=(0, 0, 0), alpha=1.0,
         accumulate=False, split_k=1, dtype=None, ksegs=None, row_map=None, rows_per_map=0, drop=None, colsum_out=None, aux_tiled=False):
    """out = epilogue(opA(A) @ opB(B)); see include/mico_hip.h (mico_gemm).  A/B are 2-D 16-bit tensors (row stride =
    leading dim).  ta: A stored [K,M]; tb: B stored [K,N].  aux_tiled: the aux tensor is an AuxTiled-sized buffer in the MLP pair's
    private layout (mico_gemm_epilogue::aux_tiled; aux_buffer() below decides)."""
    dtype = dtype or A.dtype
    if A.dtype != B.dtype:   # the kernel takes ONE element type for both operands: mixed bit patterns would multiply silently
        raise MicoHipError(f"mico_gemm operands differ in dtype ({A.dtype} x {B.dtype}): a backward pass running under another "
                           "runtime.precision than its forward?")
    if M is None:
        M = A.shape[1] if ta else A.shape[0]
    if K is None:
        K = A.shape[0] if ta else A.shape[1]
    if N is None:
        N = B.shape[1] if tb else B.shape[0]
    e = GemmEpilogue()
    e.bias = _p(bias)
    e.aux_out = _p(aux_out)
    e.aux_in = _p(aux_in)
    aux = aux_out if aux_out is not None else aux_in
    e.ldaux = aux.stride(0) if aux is not None else 0
    if aux_out is not None and aux_in is not None:
        # GELU_GRAD with both (the pre-activation-keeping MLP pair): aux_in = the pre-activation (tiled with aux_tiled), aux_out = gelu(aux_in), always
        # row-major; one leading dimension serves both
        if act != ACT_GELU_GRAD or (not aux_tiled and aux_in.stride(0) != aux_out.stride(0)):
            raise MicoHipError("mico_gemm: aux_in together with aux_out is the GELU_GRAD launch that also writes gelu(aux_in); both row-major tensors share ldaux")
    if aux_tiled:
        _check_tiled(aux_in if aux_in is not None else aux, M, N)
        e.aux_tiled = 1
        if aux_out is None or aux_in is None:
            e.ldaux = N
    e.act = act
    e.row_scale = _p(row_scale)
    e.rows_per_scale = rows_per_scale
    e.resid = _p(resid)
    e.pos = _p(pos)
    e.pos_rows = pos_rows
    e.remap_group, e.remap_skip, e.remap_offset = remap
    e.alpha = alpha
    e.accumulate = 1 if accumulate else 0
    e.row_map = _p(row_map)
    e.rows_per_map = rows_per_map
    if drop is not None:   # (p, seed, site)
        e.drop_p, e.drop_seed, e.drop_site = float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2])
    e.colsum_out = _p(colsum_out)
    if SPLITK_SLABS and accumulate and out.dtype == torch.float32 and split_k != 1 and ta and tb:
        # scratch for the split-K slabs of the weight-gradient kernels (mico_gemm_epilogue::splitk_ws): room for 8 splits of the towers'
        # large layers, 32 of BERT's small ones
        slab = M * N * 4
        ws = _splitk_scratch(min(32, max(2, SPLITK_WS_CAP // slab)) * slab, out.device)
        e.splitk_ws, e.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
    if ksegs is not None:   # (kseg, a_offsets, b_offsets)
        e.kseg, e.nseg = ksegs[0], len(ksegs[1])
        for i, (ao, bo) in enumerate(zip(ksegs[1], ksegs[2])):
            e.a_seg_off[i], e.b_seg_off[i] = ao, bo
        K = e.kseg * e.nseg
    timer = GEMM_TIMER
    timed = timer is not None and (int(ta), int(tb)) in timer.variants
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().mico_gemm(int(ta), int(tb), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0),
                              dt_code(out.dtype), C.byref(e), split_k, dt_code(dtype), _st())
    if timed:
        e1.record()
        timer.records.append(((int(ta), int(tb), int(_lib.lib().mico_gemm_last_kernel())), 2.0 * M * N * K, e0, e1))
        sig = "".join(c for c, on in (("b", bias is not None), ("x", aux_out is not None), ("i", aux_in is not None), ("g", act != ACT_NONE),
                                      ("s", row_scale is not None), ("r", resid is not None), ("m", row_map is not None),
                                      ("a", accumulate), ("d", drop is not None), ("p", pos is not None)) if on)
        timer.detail.append((M, N, K, sig + ("/f32" if out.dtype == torch.float32 else ""), split_k))
    check(rc, "mico_gemm")
    return out


def _check_tiled(aux, M, N):
    need = ((M + 255) // 256) * 256 * N
    if aux is None or not aux.is_contiguous() or aux.numel() < need:
        raise MicoHipError(f"aux_tiled: the aux tensor of an [{M}, {N}] launch is a contiguous buffer of >= {need} elements (ops.aux_buffer)")


def aux_buffer(M, N, K, dtype, device, tiled_ok=True):
    """The tensor that carries gelu'(pre-activation) from an MLP's forward GEMM (GELU_SAVE_DERIV, aux_out) to its backward (MUL_AUX, aux_in):
    (buffer, tiled).  tiled = both launches - [M, N] over a reduction of K - run on the persistent 8-phase kernel, which keeps the tensor in its
    accumulator layout (mico_gemm_epilogue::aux_tiled; M rounded up to 256 rows); otherwise (or with tiled_ok = False: launches the caller
    knows to go elsewhere - fp8 mode, three k-segments) the documented row-major [M, N]."""
    n = int(_lib.lib().mico_gemm_aux_tiled_elems(M, N, K)) if (AUX_TILED and tiled_ok) else 0
    if n > 0:
        return torch.empty((n // N, N), dtype=dtype, device=device), True
    return torch.empty((M, N), dtype=dtype, device=device), False


AUX_TILED = os.environ.get("MICO_AUX_TILED", "1") != "0"     # A/B switch: 0 keeps every aux tensor row-major


def _epilogue(bias=None, aux_out=None, aux_in=None, act=ACT_NONE, row_scale=None, rows_per_scale=0, resid=None, pos=None, pos_rows=0,
              remap=(0, 0, 0), alpha=1.0, accumulate=False, row_map=None, rows_per_map=0, drop=None, aux_tiled=False, tiled_shape=None):
    e = GemmEpilogue()
    e.bias, e.aux_out, e.aux_in = _p(bias), _p(aux_out), _p(aux_in)
    aux = aux_out if aux_out is not None else aux_in
    e.ldaux = aux.stride(0) if aux is not None else 0
    if aux_tiled:
        _check_tiled(aux, *tiled_shape)
        e.aux_tiled, e.ldaux = 1, tiled_shape[1]
    e.act = act
    e.row_scale, e.rows_per_scale = _p(row_scale), rows_per_scale
    e.resid, e.pos, e.pos_rows = _p(resid), _p(pos), pos_rows
    e.remap_group, e.remap_skip, e.remap_offset = remap
    e.alpha = alpha
    e.accumulate = 1 if accumulate else 0
    e.row_map, e.rows_per_map = _p(row_map), rows_per_map
    if drop is not None:
        e.drop_p, e.drop_seed, e.drop_site = float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2])
    return e


class Mx8:
    """An MX-fp8 operand: q uint8 [rows, K] (OCP e4m3 bytes) + scales int32 [K / 128, rows] (four E8M0 bytes per word), see mico_quant_mx8."""
    __slots__ = ("q", "scales")

    def __init__(self, q, scales):
        self.q, self.scales = q, scales

    def dequant(self):
        """fp32 [rows, K] - tests and error measurements only."""
        rows, K = self.q.shape
        v = self.q.view(torch.float8_e4m3fn).float()
        w = self.scales.t().contiguous().view(torch.uint8).view(rows, K // 128, 4).reshape(rows, K // 32).float()
        return v * torch.exp2(w - 127.0).repeat_interleave(32, dim=1)


def quant_mx8(x, pre_scale=1.0, rows=None):
    """16-bit [rows, K] (row stride = ld) -> Mx8; K % 128 == 0."""
    rows = rows if rows is not None else x.shape[0]
    K = x.shape[1]
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty((K // 128, rows), dtype=torch.int32, device=x.device)
    check(_lib.lib().mico_quant_mx8(_p(x), x.stride(0), rows, K, _p(q), q.stride(0), _p(sc), float(pre_scale), dt_code(x.dtype), _st()),
          "mico_quant_mx8")
    return Mx8(q, sc)


def gemm_mx8(A, B, out, *, dtype, M=None, **epi):
    """out = epilogue(A B^T) on the block-scaled fp8 MFMA; A [M, K] / B [N, K] Mx8 operands; dtype = the 16-bit type of a 16-bit out /
    aux tensors; epilogue keywords as ops.gemm (no split-K, no k-segments)."""
    M = M if M is not None else A.q.shape[0]
    N, K = B.q.shape
    e = _epilogue(tiled_shape=(M, N), **epi)
    timer = GEMM_TIMER
    timed = timer is not None and (0, 0) in timer.variants
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _lib.lib().mico_gemm_mx8(M, N, K, _p(A.q), A.q.stride(0), _p(A.scales), _p(B.q), B.q.stride(0), _p(B.scales), _p(out), out.stride(0),
                                  dt_code(out.dtype), C.byref(e), dt_code(dtype), _st())
    if timed:
        e1.record()
        timer.records.append(((0, 0, 4), 2.0 * M * N * K, e0, e1))
        timer.detail.append((M, N, K, "mx8" + ("/f32" if out.dtype == torch.float32 else ""), 1))
    check(rc, "mico_gemm_mx8")
    return out


def layernorm_fwd(x, gamma, beta, eps, *, out16=None, out32=None, mean=None, rstd=None, post_add=None,
                  post_rows_per_group=0, post_groups=0, split16=False, dtype=torch.float16, frame_map=None,
                  rows_per_frame=0, x_copy=None, xhat16=None, x_normalized=False, drop=None, valid_cols=0, mx8=None):
    """split16: out16 is [rows, 2*cols] and receives [hi | lo] (split-precision GEMM operand).  frame_map (int32 [frames]):
    compacting gather of whole frames out of x; the number of rows is then len(frame_map) * rows_per_frame.
    xhat16 (fp16 [rows, cols]): receives the normalised gathered rows - the half-size replacement of x_copy for the backward;
    x_normalized: x IS such a buffer (y = x gamma + beta, no statistics).  mx8: an Mx8 whose q / scales receive the fp8 operand."""
    rows, cols = x.shape[0], x.shape[1]
    if frame_map is not None:
        rows = frame_map.shape[0] * rows_per_frame
    if xhat16 is not None:
        assert xhat16.dtype == torch.float16 and xhat16.is_contiguous() and tuple(xhat16.shape) == (rows, cols)
    if x_normalized:
        assert x.dtype == torch.float16 and x.is_contiguous()
    p = _lib.LnFwdParams(x=_p(x), x_dtype=dt_code(x.dtype), x_normalized=int(bool(x_normalized)), gamma=_p(gamma), beta=_p(beta), y16=_p(out16),
                         y32=_p(out32), mean=_p(mean), rstd=_p(rstd), rows=rows, cols=cols, eps=eps, post_add=_p(post_add),
                         post_rows_per_group=post_rows_per_group, post_groups=post_groups, y16_split=int(split16), frame_map=_p(frame_map),
                         rows_per_frame=rows_per_frame, x_copy=_p(x_copy), xhat16=_p(xhat16), drop_p=float(drop[0]) if drop else 0.0,
                         drop_seed=(int(drop[1]) & 0xFFFFFFFF) if drop else 0, drop_site=int(drop[2]) if drop else 0, valid_cols=int(valid_cols),
                         q8=_p(mx8.q) if mx8 is not None else None, ldq=mx8.q.stride(0) if mx8 is not None else 0,
                         scales=_p(mx8.scales) if mx8 is not None else None)
    check(_lib.lib().mico_layernorm_fwd(C.byref(p), dt_code(dtype), _st()), "mico_layernorm_fwd")


def layernorm_fwd_mx8(x, gamma, beta, eps, *, out16, mean, rstd, dtype, frame_map=None, rows_per_frame=0, x_copy=None, xhat16=None,
                      x_normalized=False):
    """layernorm_fwd whose 16-bit output is also quantised to the MX fp8 operand of gemm_mx8 (returns the Mx8; == quant_mx8(out16))."""
    rows, cols = x.shape[0], x.shape[1]
    if frame_map is not None:
        rows = frame_map.shape[0] * rows_per_frame
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty((cols // 128, rows), dtype=torch.int32, device=x.device)
    mx = Mx8(q, sc)
    layernorm_fwd(x, gamma, beta, eps, out16=out16, mean=mean, rstd=rstd, dtype=dtype, frame_map=frame_map, rows_per_frame=rows_per_frame,
                  x_copy=x_copy, xhat16=xhat16, x_normalized=x_normalized, mx8=mx)
    return mx


def layernorm_bwd(dy, x, gamma, mean, rstd, *, dy_scale=1.0, dx_add=None, dx32=None, dx16=None, scale16=1.0, dgamma=None, dbeta=None,
                  grad_scale=1.0, dtype=torch.float16, frame_map=None, rows_per_frame=0, valid_cols=0, dx16_dst=None, dx16_frame_scale=None,
                  dx16_drop=None, x_normalized=False):
    """frame_map: dx_add / dx32 are the full stream, addressed through the frame scatter; dy / x / mean / rstd are compact.
    dx16_dst / dx16_frame_scale: dx16 is laid out for the next consumer's frame set (mico_layernorm_bwd in include/mico_hip.h).
    dx16_drop: (p, seed, site) - dx16 also carries that dropout's mask (the gradient side of a hidden-state dropout).
    x_normalized: x is the forward's fp16 xhat16 copy (mean may be None)."""
    dp, dseed, dsite = (float(dx16_drop[0]), int(dx16_drop[1]) & 0xFFFFFFFF, int(dx16_drop[2])) if dx16_drop is not None else (0.0, 0, 0)
    rows, cols = x.shape[0], x.shape[1]
    ws = None
    if dgamma is not None or dbeta is not None:
        nblk = _lib.lib().mico_layernorm_bwd_nblk(rows)
        ws = torch.empty(2 * nblk * cols, dtype=torch.float32, device=x.device)
    if x_normalized:
        assert x.dtype == torch.float16 and x.is_contiguous()
    p = _lib.LnBwdParams(dy=_p(dy), dy_dtype=dt_code(dy.dtype), dy_scale=dy_scale, x=_p(x), x_dtype=dt_code(x.dtype),
                         x_normalized=int(bool(x_normalized)), gamma=_p(gamma), mean=_p(mean), rstd=_p(rstd), dx_add=_p(dx_add), dx32=_p(dx32),
                         dx16=_p(dx16), scale16=scale16, dgamma=_p(dgamma), dbeta=_p(dbeta), grad_scale=grad_scale, ws=_p(ws), rows=rows,
                         cols=cols, frame_map=_p(frame_map), rows_per_frame=rows_per_frame, valid_cols=int(valid_cols), dx16_dst=_p(dx16_dst),
                         dx16_frame_scale=_p(dx16_frame_scale), dx16_drop_p=dp, dx16_drop_seed=dseed, dx16_drop_site=dsite)
    check(_lib.lib().mico_layernorm_bwd(C.byref(p), dt_code(dtype), _st()), "mico_layernorm_bwd")


def dropout_(x, drop):
    """In-place dropout of a 2-D fp32 / 16-bit tensor with the (p, seed, site) counter hash (see mico_dropout)."""
    rows, cols = x.shape
    check(_lib.lib().mico_dropout(_p(x), dt_code(x.dtype), rows, cols, x.stride(0), float(drop[0]), int(drop[1]) & 0xFFFFFFFF,
                                  int(drop[2]), _st()), "mico_dropout")
    return x


ATTN_SMALLQ_MAX = 80   # query rows of the one-pass short-query backward (SqCfg::QMAX in csrc/attention.hip; hd 64 only)


def attn_bwd_smallq_ok(B, H, Sq, Sk, hd, drop=None, batch0=0):
    """True when mico_attn_bwd takes its one-pass short-query kernel for this launch - the only kernel that implements
    mico_attn_params.dkv_accumulate.  The library's own condition (csrc/attention.hip, mico_attn_bwd), restated so that a caller that wants
    the two-launch ITM triplet backward can fall back to the one-launch form instead of running into the library's check."""
    import os
    if os.environ.get("MICO_ATTN_NOSMALLQ") is not None or hd != 64 or Sq > ATTN_SMALLQ_MAX:
        return False
    p = float(drop[0]) if drop else 0.0
    return p <= 0.0 or (B + batch0) * H * Sq * Sk <= 0xFFFFFFFF


def _attn_params(B, H, Sq, Sk, hd, q, k, v, o, scale, mask, q_strides, k_strides, v_strides, o_strides, drop=None, kv_batch_mod=0, batch0=0,
                 dkv_accumulate=False):
    p = AttnParams()
    p.kv_batch_mod = int(kv_batch_mod)
    p.batch0, p.dkv_accumulate = int(batch0), int(bool(dkv_accumulate))
    if drop is not None:
        p.drop_p, p.drop_seed, p.drop_site = float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2])
    p.B, p.H, p.Sq, p.Sk, p.hd = B, H, Sq, Sk, hd
    p.q_bs, p.q_rs = q_strides
    p.k_bs, p.k_rs = k_strides
    p.v_bs, p.v_rs = v_strides
    p.o_bs, p.o_rs = o_strides
    p.scale = scale
    p.mask = _p(mask)
    if mask is None:
        p.mask_mode = 0
    elif mask.dim() == 2:
        p.mask_mode = 1
    else:
        p.mask_mode = 2
    return p


def attn_fwd(q, k, v, o, lse, *, B, H, Sq, Sk, hd, scale, mask=None, q_strides, k_strides, v_strides, o_strides, drop=None,
             kv_batch_mod=0):
    """q/k/v/o are 16-bit tensors (possibly views into one fused projection buffer); *_strides = (batch, row) in
    elements.  mask: additive fp32 [B,Sk] or [B,Sq,Sk]."""
    p = _attn_params(B, H, Sq, Sk, hd, q, k, v, o, scale, mask, q_strides, k_strides, v_strides, o_strides, drop, kv_batch_mod)
    rc = _lib.lib().mico_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), C.byref(p), dt_code(q.dtype), _st())
    check(rc, "mico_attn_fwd")


def attn_bwd(q, k, v, o, do, lse, dq, dk, dv, delta, *, B, H, Sq, Sk, hd, scale, mask=None, q_strides, k_strides,
             v_strides, o_strides, drop=None, kv_batch_mod=0, batch0=0, dkv_accumulate=False):
    """kv_batch_mod > 0: k / v hold kv_batch_mod batch entries shared modulo (see mico_attn_params); dk / dv are [B, ...] as always.
    batch0: this launch covers entries batch0 .. batch0 + B of a larger batch (dropout counters); dkv_accumulate: dk / dv += (short-query kernel)."""
    p = _attn_params(B, H, Sq, Sk, hd, q, k, v, o, scale, mask, q_strides, k_strides, v_strides, o_strides, drop, kv_batch_mod, batch0, dkv_accumulate)
    rc = _lib.lib().mico_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(dq), _p(dk), _p(dv), _p(delta),
                                  C.byref(p), dt_code(q.dtype), _st())
    check(rc, "mico_attn_bwd")


def rope(x, bs, rs, B, N, H, hd, cos_t, sin_t, inverse=False):
    check(_lib.lib().mico_rope(_p(x), bs, rs, B, N, H, hd, _p(cos_t), _p(sin_t), int(inverse), dt_code(x.dtype), _st()),
          "mico_rope")


def im2row(pixels, rows16, P, kpad):
    B, Cc, H, W = pixels.shape
    check(_lib.lib().mico_im2row(_p(pixels), _p(rows16), B, Cc, H, W, P, kpad, dt_code(rows16.dtype), _st()), "mico_im2row")


def cast_f32_to_16(src, dst, *, cols=None, cols_pad=None, scale=1.0):
    rows = src.shape[0]
    cols = cols if cols is not None else src.shape[1]
    cols_pad = cols_pad if cols_pad is not None else dst.shape[1]
    check(_lib.lib().mico_cast_f32_to_16(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, cols_pad, scale,
                                         dt_code(dst.dtype), _st()), "mico_cast_f32_to_16")
    return dst


def cast_16_to_f32(src, dst, *, scale=1.0, accumulate=False):
    rows, cols = src.shape
    check(_lib.lib().mico_cast_16_to_f32(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, scale,
                                         int(accumulate), dt_code(src.dtype), _st()), "mico_cast_16_to_f32")
    return dst


def gather_rows_cast(src, dst, *, remap=(0, 0, 0), row_scale=None, rows_per_scale=0, scale=1.0, frame_map=None,
                     rows_per_frame=0, dst_map=None):
    """dst_map: the frames listed in frame_map land in the frame slots dst_map[j] of dst (only those rows are written)."""
    rows, cols = dst.shape
    if dst_map is not None:
        rows = frame_map.numel() * rows_per_frame
    check(_lib.lib().mico_gather_rows_cast(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, remap[0], remap[1],
                                           remap[2], _p(row_scale), rows_per_scale, scale, _p(frame_map), rows_per_frame,
                                           _p(dst_map), dt_code(dst.dtype), _st()),
          "mico_gather_rows_cast")
    return dst


def colsum(x, out, *, rows=None, cols=None, ld=None, scale=1.0, accumulate=False):
    rows = rows if rows is not None else x.shape[0]
    cols = cols if cols is not None else x.shape[1]
    ld = ld if ld is not None else x.stride(0)
    check(_lib.lib().mico_colsum(_p(x), dt_code(x.dtype), ld, rows, cols, _p(out), scale, int(accumulate), _st()),
          "mico_colsum")
    return out


def dw_colfold(dwt, dbt, gamma, beta, dw, db=None):
    """dw[m, n] += dwt[m, n] * gamma[n] + dbt[m] * beta[n];  db[m] += dbt[m]  (mico_dw_colfold: a LayerNorm's affine folded into the weight
    gradient of the Linear it feeds - dwt / dbt come from the weight-gradient GEMM against the normalised rows)."""
    M, N = dwt.shape
    assert dwt.is_contiguous() and dwt.dtype == dbt.dtype == gamma.dtype == beta.dtype == dw.dtype == torch.float32 and dw.stride(1) == 1
    assert dw.shape[0] >= M and dw.shape[1] == N and dbt.numel() >= M and gamma.numel() == N and beta.numel() == N
    check(_lib.lib().mico_dw_colfold(_p(dwt), _p(dbt), _p(gamma), _p(beta), _p(dw), dw.stride(0), _p(db), M, N, _st()), "mico_dw_colfold")


def cls_rows(x, B, group_rows, cls, pos0):
    check(_lib.lib().mico_cls_rows(_p(x), x.stride(0), B, group_rows, _p(cls), _p(pos0), x.shape[1], _st()), "mico_cls_rows")


def add_f32(a, b, y=None, y16=None, scale16=1.0, dtype=torch.float16):
    check(_lib.lib().mico_add_f32(_p(a), _p(b), _p(y), _p(y16), a.numel(), scale16,
                                  dt_code(y16.dtype if y16 is not None else dtype), _st()), "mico_add_f32")


def swiglu_fwd(x1, x2, h):
    check(_lib.lib().mico_swiglu_fwd(_p(x1), _p(x2), _p(h), x1.numel(), dt_code(x1.dtype), _st()), "mico_swiglu_fwd")


def swiglu_fwd_f32(x1, x2, h):
    check(_lib.lib().mico_swiglu_fwd_f32(_p(x1), _p(x2), _p(h), x1.numel(), _st()), "mico_swiglu_fwd_f32")


def swiglu_bwd(x1, x2, dh, dx1, dx2):
    check(_lib.lib().mico_swiglu_bwd(_p(x1), _p(x2), _p(dh), _p(dx1), _p(dx2), x1.numel(), dt_code(x1.dtype), _st()),
          "mico_swiglu_bwd")


def bert_embed_fwd(ids, word, pos, type0, out, S):
    rows, cols = out.shape
    check(_lib.lib().mico_bert_embed_fwd(_p(ids), _p(word), _p(pos), _p(type0), _p(out), rows, S, cols, word.shape[0], _st()),
          "mico_bert_embed_fwd")


def embed_scatter_add(ids, dsum, dword, dpos, dtype0, S, scale=1.0):
    rows, cols = dsum.shape
    vocab = dword.shape[0] if dword is not None else 0
    check(_lib.lib().mico_embed_scatter_add(_p(ids), _p(dsum), _p(dword), _p(dpos), _p(dtype0), rows, S, cols, vocab, scale,
                                            _st()), "mico_embed_scatter_add")


def ce_fwd_bwd(logits, target, *, cols=None, ignore_index=-100, label_smoothing=0.0, logits_scale=1.0, row_loss=None,
               row_lse=None, dlogits=None, dscale_ptr=None, dscale=1.0):
    rows = logits.shape[0]
    cols = cols if cols is not None else logits.shape[1]
    check(_lib.lib().mico_ce_fwd_bwd(_p(logits), dt_code(logits.dtype), logits.stride(0), rows, cols, _p(target),
                                     ignore_index, label_smoothing, logits_scale, _p(row_loss), _p(row_lse), _p(dlogits),
                                     dt_code(dlogits.dtype) if dlogits is not None else 0,
                                     dlogits.stride(0) if dlogits is not None else 0, _p(dscale_ptr), dscale, 0, _st()),
          "mico_ce_fwd_bwd")


def token_mask(tokens, mask_prob, u_mask, u_kind, u_tok, mask_token, range_start, range_end):
    """(masked token ids, labels) of the caption loss's TokenMasker on the device; see mico_token_mask.  u_mask [rounds, rows, S]."""
    rows, S = tokens.shape
    toks = tokens.contiguous()
    out, labels = torch.empty_like(toks), torch.empty_like(toks)
    check(_lib.lib().mico_token_mask(_p(toks), rows, S, float(mask_prob), _p(u_mask.contiguous()), u_mask.shape[0], _p(u_kind.contiguous()),
                                     _p(u_tok.contiguous()), int(mask_token), int(range_start), int(range_end), _p(out), _p(labels), _st()),
          "mico_token_mask")
    return out, labels


def itm_sample(sim, diag_offset, u):
    """One hard-negative index per row of the fp32 similarity logits `sim` [rows, cols]; see mico_itm_sample."""
    sim = sim.contiguous()
    out = torch.empty(sim.shape[0], dtype=torch.int64, device=sim.device)
    check(_lib.lib().mico_itm_sample(_p(sim), sim.stride(0), sim.shape[0], sim.shape[1], int(diag_offset), _p(u.contiguous()), _p(out), _st()),
          "mico_itm_sample")
    return out


def win_attn_fwd(qkv, out, lse, bias_table, batch, res, heads, shift, scale):
    """Swin (shifted-)window attention in token order; see mico_win_attn_fwd."""
    check(_lib.lib().mico_win_attn_fwd(_p(qkv), _p(out), _p(lse), _p(bias_table), batch, res, heads, shift, float(scale), dt_code(qkv.dtype),
                                       _st()), "mico_win_attn_fwd")


def win_attn_bwd(qkv, dout, lse, bias_table, dqkv, dbias_table, batch, res, heads, shift, scale, dbias_scale=1.0):
    check(_lib.lib().mico_win_attn_bwd(_p(qkv), _p(dout), _p(lse), _p(bias_table), _p(dqkv), _p(dbias_table), batch, res, heads, shift,
                                       float(scale), float(dbias_scale), dt_code(qkv.dtype), _st()), "mico_win_attn_bwd")


def patch_merge(x, out, batch, res, channels, backward=False):
    """PatchMerging's 2x2 gather on the fp32 stream ([batch, res*res, C] -> [batch, res*res/4, 4C]) or its inverse (backward)."""
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    check(_lib.lib().mico_patch_merge(_p(x), _p(out), batch, res, channels, int(bool(backward)), _st()), "mico_patch_merge")


def l2norm_fwd(x, y, inv_norm):
    check(_lib.lib().mico_l2norm_fwd(_p(x), _p(y), _p(inv_norm), x.shape[0], x.shape[1], _st()), "mico_l2norm_fwd")


def l2norm_bwd(dy, y, inv_norm, dx):
    check(_lib.lib().mico_l2norm_bwd(_p(dy), _p(y), _p(inv_norm), _p(dx), y.shape[0], y.shape[1], _st()), "mico_l2norm_bwd")


def sgemm(A, B, out, *, ta=False, tb=False, M=None, N=None, K=None, alpha=1.0, beta=0.0, bias=None):
    """Exact fp32 small GEMM (heads / similarity matrices)."""
    if M is None:
        M = A.shape[1] if ta else A.shape[0]
    if K is None:
        K = A.shape[0] if ta else A.shape[1]
    if N is None:
        N = B.shape[1] if tb else B.shape[0]
    check(_lib.lib().mico_sgemm_small(int(ta), int(tb), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(out),
                                      out.stride(0), alpha, beta, _p(bias), _st()), "mico_sgemm_small")
    return out


def gelu_f32(x, y):
    check(_lib.lib().mico_gelu_f32(_p(x), _p(y), x.numel(), _st()), "mico_gelu_f32")


def gelu_bwd_f32(x, dy, dx):
    check(_lib.lib().mico_gelu_bwd_f32(_p(x), _p(dy), _p(dx), x.numel(), _st()), "mico_gelu_bwd_f32")


def gelu_16(x, y):
    check(_lib.lib().mico_gelu_16(_p(x), _p(y), x.numel(), dt_code(x.dtype), _st()), "mico_gelu_16")


def gelu_bwd_16(x, dy, dx):
    check(_lib.lib().mico_gelu_bwd_16(_p(x), _p(dy), _p(dx), x.numel(), dt_code(x.dtype), _st()), "mico_gelu_bwd_16")


def cls_pool_fwd(tokens, pooled, b, n, frame_stride, D):
    check(_lib.lib().mico_cls_pool_fwd(_p(tokens), _p(pooled), b, n, frame_stride, D, _st()), "mico_cls_pool_fwd")


def cls_pool_bwd(dpooled, dtokens, b, n, frame_stride, D):
    check(_lib.lib().mico_cls_pool_bwd(_p(dpooled), _p(dtokens), b, n, frame_stride, D, _st()), "mico_cls_pool_bwd")


def pool_video_fwd(tokens, pooled, frames, N, D):
    check(_lib.lib().mico_pool_video_fwd(_p(tokens), _p(pooled), frames, N, D, _st()), "mico_pool_video_fwd")


def pool_video_bwd(dpooled, dtokens, frames, N, D):
    check(_lib.lib().mico_pool_video_bwd(_p(dpooled), _p(dtokens), frames, N, D, _st()), "mico_pool_video_bwd")
