"""Execution settings and the 16-bit weight cache of the MI355X engine.

compute dtype: torch.float16 is the parity configuration (embeddings/logits within 1e-3 of the fp32 reference),
torch.bfloat16 the throughput configuration BASELINE.json's metric is quoted in.  Parameters stay fp32 (state-dict
compatible with the reference); a 16-bit copy of every GEMM weight is (re)materialised only when the parameter's
version counter changes, i.e. once per optimiser step.
"""
import contextlib
import os
import weakref

import torch

from . import ops


class _Cfg:
    compute_dtype = torch.bfloat16
    # internal 16-bit gradients are carried multiplied by grad_scale (fp16 only: guards against underflow); every
    # function de-scales what it hands back, so autograd sees true-scale fp32 gradients.
    grad_scale = {torch.float16: 4096.0, torch.bfloat16: 1.0}
    # fp16 parity configuration: forward GEMMs use hi/lo split weights (x W_hi + x W_lo in one launch through the kernel's
    # k-segments), which removes the weight-rounding half of the fp16 GEMM error.  Off for bf16 (throughput configuration).
    split_fp16 = True
    # what is split when split_fp16 is on: "full" = weights AND LayerNorm outputs (3 k-segments, x_hi W_hi + x_lo W_hi + x_hi W_lo),
    # "weights" = weights only (2 k-segments, x W_hi + x W_lo: removes the weight-rounding half of the error at 2x the MFMA work)
    split_mode = "full"
    # the tower backward's LayerNorm backward writes the NEXT branch's 16-bit gradient operand itself (functional._tower_backward: handover);
    # MICO_NO_GRAD_HANDOVER: A/B switch, every branch gathers its operand with mico_gather_rows_cast
    fuse_grad_handover = os.environ.get("MICO_NO_GRAD_HANDOVER") is None
    ln_grad_16bit = os.environ.get("MICO_LN_GRAD_FP32") is None   # gradient at the LayerNorm outputs stored 16-bit (functional._tower_backward)
    # activation diet level 3: qkv / fc1 weight gradients taken against the kept fp16 NORMALISED rows, the LayerNorm's gamma / beta applied to the
    # [out, in] result (ops.dw_colfold) instead of re-creating the LayerNorm output over all rows first (functional._tower_backward)
    ln_fold_wgrad = os.environ.get("MICO_LN_NOFOLD") is None
    # shared cross-attention K/V memory (functional.CrossKVFn) laid out [rows][layer][K | V] instead of [layer][rows][K | V]: the 12 layers' projections
    # are ONE GEMM (N = 12 x 2 D), and in the backward the condition-token gradient one K = 12 x 2 D product (no fp32 read-modify-write of the
    # [rows, D] gradient per layer) and the weight gradients one launch; the attention kernels read a layer through its row stride
    kv_interleaved = os.environ.get("MICO_KV_LAYER_MAJOR") is None
    head_split_blocks = 0     # plain fp16 only: the first n tower blocks in a split mode (see enter_block)
    head_split_mode = "weights"
    # BASELINE.json configs[4] ("fp8 MFMA"): the ViT towers' and BERT's forward and input-gradient GEMMs run on the block-scaled fp8 MFMA
    # (OCP MX e4m3, one E8M0 scale per 32 reduction elements; mico_gemm_mx8) - weight gradients, attention, LayerNorm, the residual stream
    # and every loss stay as in the 16-bit configuration of compute_dtype.  Off by default (the 16-bit path is the parity path).
    fp8 = False
    # fp8 mode: (first, last) tower blocks that stay in the 16-bit type (runtime.set_fp8_16bit_blocks)
    fp8_16bit_blocks = tuple(int(v) for v in os.environ.get("MICO_FP8_16BIT_BLOCKS", "0,0").split(","))
    # training: project the cross-attention K/V of a step's condition tokens once (functional.CrossKVFn) instead of in every BERT pass
    share_cross_kv = True
    # one gradient arena per backward pass for all the BERT passes of a step (functional.GradArena.session): no per-parameter sums by autograd
    share_grad_arena = True
    # the BERT passes that read one shared cross-attention K/V memory (ITM triplet + captioning) accumulate its 16-bit gradient in ONE buffer
    # (functional.DkvSession: the first backward writes it, later ones add in the attention kernel) instead of a buffer per pass summed out of
    # place by autograd at the step's memory peak; MICO_DKV_PER_PASS=1: a buffer per pass (A/B runs)
    dkv_inplace = os.environ.get("MICO_DKV_PER_PASS") is None
    # fp8 mode: the LayerNorm that feeds qkv / fc1 writes the GEMM's block-scaled fp8 operand itself (mico_layernorm_fwd_mx8) instead of a
    # quantisation pass over its 16-bit output; MICO_FP8_NO_FUSED_QUANT=1 for A/B runs
    fp8_fused_quant = os.environ.get("MICO_FP8_NO_FUSED_QUANT") is None
    # fc1 of a tower forward that keeps no GELU' (no backward, or the activation diet recomputes the pair) runs the GELU-only epilogue
    # (half the output bytes); MICO_FC1_PAIR_ALWAYS=1: the pair epilogue everywhere (A/B runs)
    fc1_plain_gelu = os.environ.get("MICO_FC1_PAIR_ALWAYS") is None
    # a tower block that keeps its MLP intermediates keeps ONE 16-bit tensor, the fc1 pre-activation, instead of gelu and gelu' (functional.
    # _mlp_keeps_pre: half the bytes per kept block, i.e. twice the blocks without an fc1 recompute in a given budget); MICO_MLP_STASH=pair: the
    # round-5 form (A/B runs)
    mlp_keep_pre = os.environ.get("MICO_MLP_STASH", "pre") != "pair"


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


def snapshot():
    """The precision state a forward pass ran under; see using()."""
    return (CFG.compute_dtype, CFG.split_fp16, CFG.split_mode, CFG.fp8, CFG.head_split_blocks, CFG.head_split_mode)


def restore(state):
    CFG.compute_dtype, CFG.split_fp16, CFG.split_mode, CFG.fp8, CFG.head_split_blocks, CFG.head_split_mode = state


@contextlib.contextmanager
def using(state):
    """Re-establish a snapshot() for the duration of a backward pass: saved activations are in the forward's dtype / gradient scale,
    so the backward must take its weights, its gradient scale and its split mode from the same state even when it runs outside the
    `with precision(...)` block the forward ran in (fp16 activations were otherwise multiplied by bf16-bit weights)."""
    old = snapshot()
    restore(state)
    try:
        yield
    finally:
        restore(old)


def enter_block(i, base, subln_swiglu=False, depth=None):
    """Precision state of tower block i under the pass-wide state `base` (a snapshot()): with CFG.head_split_blocks = n the first n
    blocks of a plain-fp16 tower run their forward GEMMs as x W_hi + x W_lo (the weights-split mode; EVA02-style towers - RoPE + sub-LN
    + SwiGLU, subln_swiglu=True - take the 3-segment mode with the fp32 gate there: measured on the reference goldens of the depth-2
    B/16 tower, weights-split alone leaves the fused feat_vd at 1.39e-3, the 3-segment mode at 7.8e-4).  Rounding errors made in the
    first blocks pass through every later block; on the 40-block g/14 golden four such blocks take plain fp16 from 7.1e-4 / 1.0e-3
    (token rows / feat_v) to 7.4e-4 / 7.5e-4 for +2 % step time, where the same four blocks at the END of the tower change nothing
    (tools/precision_probe.py --tail).  The tower loops call this at the top of every block and restore(base) when they leave."""
    dt, split, mode, fp8, n, hmode = base
    if n and i < n and dt == torch.float16 and not split and not fp8:
        restore((dt, True, "full" if subln_swiglu else hmode, False, n, hmode))
    elif fp8 and depth is not None and (i < CFG.fp8_16bit_blocks[0] or i >= depth - CFG.fp8_16bit_blocks[1]):
        # fp8 mode with its first / last blocks kept in the 16-bit type (set_fp8_16bit_blocks): e4m3 rounding made in the first blocks passes
        # through every later block, and the last blocks' errors reach the output unattenuated - the drift / speed trade-off is measured in
        # tests/test_full_size_gpu.py::test_config5_video_caption_step
        restore((dt, split, mode, False, n, hmode))
    else:
        restore(base)


def fc1_recompute_weight(i, subln_swiglu=False):
    """Relative cost of re-running block i's fc1 GEMM under the CURRENT pass-wide precision state (what enter_block(i, snapshot()) would set):
    2 for a weights-split head block (x W_hi + x W_lo: the reduction is twice as long), 3 where activations are split as well, else 1.
    functional.tower_plan weighs the blocks' MLP intermediates with it."""
    dt, split, mode, fp8, n, hmode = snapshot()
    if n and i < n and dt == torch.float16 and not split and not fp8:
        split, mode = True, ("full" if subln_swiglu else hmode)
    if not split:
        return 1
    return 3 if mode == "full" else 2


def saved_precision(backward):
    """Decorator for autograd.Function.backward: runs it under the precision state its forward stored with remember_precision(ctx)."""
    import functools

    @functools.wraps(backward)
    def wrapped(ctx, *grads):
        with using(ctx._mico_precision):
            return backward(ctx, *grads)
    return wrapped


def remember_precision(ctx):
    ctx._mico_precision = snapshot()


_UID = [0]


def param_uid(p):
    """Process-unique id of a parameter OBJECT (stored on it).  id(p) / data_ptr / _version can all repeat once a model has been
    freed and another allocated - a cache keyed on them could hand a new model the previous model's 16-bit weights."""
    u = getattr(p, "_mico_uid", None)
    if u is None:
        _UID[0] += 1
        u = _UID[0]
        p._mico_uid = u
    return u


_W16 = {}
_COPIES = {}      # id(param) -> {cache key: (buffer, first row, kp, split, channel_sum)}: every 16-bit copy derived from it
_ENTRY_SRC = {}   # cache key -> ids of the parameters the entry was built from


def w16(key, params, build):
    """16-bit derived weight, cached on (key, dtype) and invalidated by the source parameters' version counters."""
    dt = CFG.compute_dtype
    ver = tuple((param_uid(p), p.data_ptr(), p._version) for p in params)
    hit = _W16.get((key, dt))
    if hit is not None and hit[0] == ver:
        return hit[1]
    _prune_dead()
    with torch.no_grad():
        t = build(dt)
    _W16[(key, dt)] = (ver, t)
    _ENTRY_REFS[(key, dt)] = [weakref.ref(p) for p in params]
    return t


_ENTRY_REFS = {}


def _prune_dead():
    """drop the cached copies of parameters that no longer exist (a freed model would otherwise pin GBs of 16-bit weights)"""
    dead = [k for k, refs in _ENTRY_REFS.items() if any(r() is None for r in refs)]
    for k in dead:
        _W16.pop(k, None)
        _ENTRY_REFS.pop(k, None)
        _ENTRY_SRC.pop(k, None)
    if dead:
        gone = set(dead)
        for uid in list(_COPIES):
            for k in [k for k in _COPIES[uid] if k in gone]:
                del _COPIES[uid][k]
            if not _COPIES[uid]:
                del _COPIES[uid]


def clear_weight_cache():
    _W16.clear()
    _COPIES.clear()
    _ENTRY_SRC.clear()
    _ENTRY_REFS.clear()


def _live_copies(p):
    out = []
    for key, c in _COPIES.get(param_uid(p), {}).items():
        hit = _W16.get(key)
        if key[1] == CFG.compute_dtype and hit is not None and hit[1] is c[0]:
            out.append((key, c))
    return out


def weight_mirror(p):
    """(ptr, ld, lo_off, cols, dtype code) of THE cached 16-bit copy of parameter p in the active compute dtype, for
    mico_adamw_step to refresh in the pass that updates p - or None when p has no copy, several (the patch-embed weight:
    3-channel and channel-summed forms) or only a transformed one."""
    live = _live_copies(p)
    if len(live) != 1 or live[0][1][4]:
        return None
    buf, row0, kp, split, _ = live[0][1]
    ld = buf.stride(0)
    return (buf.data_ptr() + row0 * ld * 2, ld, kp if split else 0, p.numel() // p.shape[0], ops.dt_code(buf.dtype))


def after_optimizer_step(refreshed_ids):
    """Called by mico_amd.optim.AdamW.step() with the ids of the parameters whose mirrors it refreshed in place.  A cache
    entry stays valid iff every parameter it was built from was refreshed that way (raw-pointer updates do not bump tensor
    versions, so the version check alone would keep stale copies); every other entry is dropped and rebuilt lazily.
    A foreign optimizer that writes through `.data` (as the reference's does) must call clear_weight_cache() after its step."""
    for key in list(_W16):
        src = _ENTRY_SRC.get(key)
        # only the mirrors in the ACTIVE compute dtype were refreshed (_live_copies): an entry of the other dtype built from the same
        # parameters is stale even though its parameter ids are in refreshed_ids (e.g. bf16 training interleaved with fp16 evaluation)
        if src is None or key[1] != CFG.compute_dtype or not all(i in refreshed_ids for i in src):
            del _W16[key]
            _ENTRY_SRC.pop(key, None)
            _ENTRY_REFS.pop(key, None)


def cast_weight(w, dt, k_pad=None, n_pad=None):
    """fp32 [N, K...] parameter -> 16-bit [N(_pad), K_pad] (zero padded), K flattened."""
    w2 = w.detach().reshape(w.shape[0], -1)
    N, K = w2.shape
    kp = k_pad or ops.pad8(K)
    out = torch.zeros((n_pad or N, kp), dtype=dt, device=w.device) if (n_pad and n_pad != N) else torch.empty(
        (N, kp), dtype=dt, device=w.device)
    ops.cast_f32_to_16(w2.contiguous(), out[:N], cols=K, cols_pad=kp)
    return out


def split_precision():
    return CFG.compute_dtype == torch.float16 and CFG.split_fp16


@contextlib.contextmanager
def fp8_mode(on=True):
    old = CFG.fp8
    CFG.fp8 = bool(on)
    try:
        yield
    finally:
        CFG.fp8 = old


def set_fp8_16bit_blocks(first=0, last=0):
    """In fp8 mode keep the tower's first / last blocks in the 16-bit type (accuracy against speed; (0, 0) = every block on the fp8 MFMA)."""
    CFG.fp8_16bit_blocks = (int(first), int(last))


def fp8_enabled():
    """fp8 GEMMs are an alternative to the split-precision parity configuration, never combined with it."""
    return CFG.fp8 and not split_precision()


def gemm_weight_mx8(plist, tag="w", transposed=False):
    """MX-fp8 copy (ops.Mx8) of nn.Linear-style parameters (rows of all `plist` entries concatenated): W [N, K] quantised along K for
    y = x W^T, or (transposed) W^T [K, N] quantised along N for dx = dy W - each along ITS reduction dimension, which is what the block
    scales require.  Cached per parameter version like the 16-bit copies and re-quantised after an optimizer step."""
    def build(dt):
        w = plist[0].detach() if len(plist) == 1 else torch.cat([p.detach() for p in plist], 0)
        w2 = w.reshape(w.shape[0], -1)
        if transposed:
            w2 = w2.t()
        w16 = w2.contiguous().to(dt)      # layout copy + cast of a weight, once per optimizer step (not on the per-token path)
        return ops.quant_mx8(w16)
    return w16(("mx8t" if transposed else "mx8", tag, param_uid(plist[0])), plist, build)


def split_activations():
    """LayerNorm emits [hi | lo] operands (the third k-segment of a split-precision forward GEMM)."""
    return split_precision() and CFG.split_mode == "full"


def gemm_weight(plist, tag="w", k_pad=None, n_pad=None, channel_sum=False, ln_fed=True):
    """16-bit GEMM weight for nn.Linear-style parameters (rows of all `plist` entries concatenated).
    Returns (w16, ksegs): w16 is the [N(_pad), Kp] operand (a strided view of the [N, 2 Kp] hi|lo buffer in split-precision
    mode), ksegs the k-segment descriptor for *forward* launches (None in plain mode).
    ln_fed: the GEMM's input is a LayerNorm output (qkv, fc1) - the split mode "weights-ln" splits only those weights, "weights-res" only the
    others (the projections that write the residual stream: attn.proj, fc2)."""
    split = split_precision() and (CFG.split_mode != "weights-res" if ln_fed else CFG.split_mode != "weights-ln")

    def build(dt):
        w = plist[0].detach() if len(plist) == 1 else torch.cat([p.detach() for p in plist], 0)
        if channel_sum:
            w = w.sum(1)
        w2 = w.reshape(w.shape[0], -1).float().contiguous()
        N, K = w2.shape
        kp = k_pad or ops.pad8(K)
        do_split = split and kp % 64 == 0   # k-segments need 64-element segments; every GEMM weight on the path has that
        rows = n_pad or N
        out = torch.zeros((rows, 2 * kp if do_split else kp), dtype=dt, device=w.device)
        ops.cast_f32_to_16(w2, out[:N], cols=K, cols_pad=kp)
        out._mico_split = do_split
        if do_split:
            hi32 = torch.zeros((N, kp), dtype=torch.float32, device=w.device)
            ops.cast_16_to_f32(out[:N, :kp], hi32)
            resid = torch.zeros((N, kp), dtype=torch.float32, device=w.device)
            resid[:, :K] = w2 - hi32[:, :K]
            ops.cast_f32_to_16(resid, out[:N, kp:], cols=kp, cols_pad=kp)
        key = ((tag, param_uid(plist[0]), k_pad, n_pad, split), dt)
        r0 = 0
        for p in plist:   # every source parameter occupies a row range of this buffer (see weight_mirror)
            _COPIES.setdefault(param_uid(p), {})[key] = (out, r0, kp, do_split, channel_sum)
            r0 += p.shape[0]
        _ENTRY_SRC[key] = [param_uid(p) for p in plist]
        return out

    buf = w16((tag, param_uid(plist[0]), k_pad, n_pad, split), plist, build)
    if getattr(buf, "_mico_split", False):
        kp = buf.shape[1] // 2
        return buf[:, :kp], (kp, [0, 0], [0, kp])
    return buf, None


_tower_chunk = None
step_staged = False      # the current step differentiates its BERT passes inside the forward (MiCo.forward(backward_scale=...)); read by functional.tower_plan


def tower_chunk_override():
    """Frames per ViT tower pass forced by set_tower_chunk (None = size it from the free device memory)."""
    return _tower_chunk


def set_tower_chunk(frames):
    global _tower_chunk
    _tower_chunk = int(frames) if frames else None


_activation_diet = None
last_tower_plan = None      # dict(frames=, frames_per_pass=, diet=, mlp_blocks_kept=, rows_fp16_normalised=, kept_fraction=[, oom_retry=]) of the most recent training-mode tower forward


def reset():
    """Release what the engine caches between steps outside the parameters: split-K scratch buffers, 16-bit / fp8 weight copies, shared
    gradient arenas of finished backward passes."""
    ops.release_scratch()
    clear_weight_cache()
    from . import functional
    functional.GradArena._shared.clear()


def activation_diet_override():
    """Saved-activation level of the ViT tower forced by set_activation_diet (None = the cheapest level that fits the free memory)."""
    return _activation_diet


def set_activation_diet(level, mlp_blocks=0):
    """0: keep everything the backward reads; 1: drop the two MLP intermediates (GELU output and GELU', 4 * hidden of the 20 D + 4 hidden
    bytes per token and block) and recompute them in the backward with one fc1 GEMM; 2: also drop the LayerNorm outputs (recomputed from
    the saved fp32 rows); 3: as 2 with the LayerNorm input rows kept as fp16 normalised rows instead of fp32 copies, and the MLP
    intermediates of the LAST `mlp_blocks` blocks kept (functional.TowerDiet).  None: automatic (functional.tower_plan).
    activation_diet_override() -> None | (level, mlp_blocks)."""
    global _activation_diet
    _activation_diet = None if level is None else (int(level), int(mlp_blocks))


_grad_slice_hook = None


def set_grad_slice_hook(fn):
    """fn(flat_slice, params) or None.  The ViT tower's backward calls it with a contiguous slice of its gradient arena as soon as
    the gradients of a block are final, long before autograd hands the (views of the) gradients to the parameters: the
    data-parallel reducer starts that block's all-reduce there, overlapped with the rest of the backward."""
    global _grad_slice_hook
    _grad_slice_hook = fn


def grad_slice_hook():
    return _grad_slice_hook


_MEM_TRACE = os.environ.get("MICO_MEM_TRACE") is not None


def mem_trace(tag):
    """MICO_MEM_TRACE=1: print the device memory in use at a stage boundary of the step (syncs: a debugging aid, never on in a timed run)."""
    if _MEM_TRACE and torch.cuda.is_available():
        import sys
        torch.cuda.synchronize()
        print(f"[mem] {tag:28s} allocated {torch.cuda.memory_allocated() / 2**30:7.1f} GB   peak {torch.cuda.max_memory_allocated() / 2**30:7.1f} GB   "
              f"reserved {torch.cuda.memory_reserved() / 2**30:7.1f} GB", file=sys.stderr, flush=True)
