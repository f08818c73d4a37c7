"""Autograd functions of the MI355X engine: hand-written forward AND backward schedules over libmico_hip.so kernels.

Granularity is one Function per tower (EVA ViT, BERT) plus a few small ones (linear, layer-norm, losses): PyTorch's
autograd only routes gradients between them and into the fp32 nn.Parameters; every FLOP inside runs in the HIP kernels
(mico_amd/ops.py).  Residual streams and their gradients are fp32 buffers updated in place by GEMM epilogues; GEMM
operands are 16-bit (runtime.compute_dtype()).
"""
import math
import os

import logging
import torch

from . import ops, runtime

_log = logging.getLogger("mico_amd")


def _split_k(m_out, n_out, k_red):
    """Split factor for the long-reduction weight-gradient GEMMs: 512 workgroup slots (256 CUs x 2) must be filled in whole
    waves - e.g. 48 x 11 = 528 tiles unsplit would run as one full wave plus a 16-block tail at 2x the time.  Cost model:
    waves(s) * (k-tiles per split + fixed prologue / atomic-epilogue cost in k-tile units)."""
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    ktiles = (k_red + 63) // 64
    best, best_cost = 1, None
    for s in range(1, 33):
        if s > 1 and ktiles // s < 16:
            break
        waves = -(-(tiles * s) // 512)
        cost = waves * (-(-ktiles // s) + 12)
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


class GradArena:
    """fp32 parameter-gradient storage of one backward pass: ONE zero-filled flat buffer with a view per parameter (a ViT-g
    backward otherwise issues ~1600 tiny fill launches per step).  Parameters nobody asked a view for report None.
    groups: lists of parameter positions whose views must be ADJACENT in the given order (BERT's query | key | value weights: the
    fused [3 D, D] weight-gradient GEMM then accumulates straight into them, fused())."""

    def __init__(self, params, groups=None):
        self.shapes = [tuple(p.shape) for p in params]
        self.offsets, n = [None] * len(params), 0
        lead = {g[0]: g for g in (groups or [])}
        for i, p in enumerate(params):
            if self.offsets[i] is not None:
                continue
            for j in lead.get(i, [i]):
                assert self.offsets[j] is None and (j == i or j > i), "a group follows its first member"
                self.offsets[j] = n
                n += (params[j].numel() + 3) // 4 * 4          # keep every view 16-byte aligned
        self.flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        self.views = [None] * len(params)

    def get(self, i):
        if self.views[i] is None:
            o = self.offsets[i]
            numel = 1
            for d in self.shapes[i]:
                numel *= d
            self.views[i] = self.flat[o:o + numel].view(self.shapes[i])
        return self.views[i]

    def fused(self, idxs, shape):
        """one view over the adjacent parameters idxs (a group of the constructor), shaped `shape`"""
        n = 0
        for a, b in zip(idxs, idxs[1:]):
            assert self.offsets[b] == self.offsets[a] + self.get(a).numel(), "not adjacent (numel % 4 != 0 or not a group)"
        for i in idxs:
            n += self.get(i).numel()
        return self.flat[self.offsets[idxs[0]]:self.offsets[idxs[0]] + n].view(shape)

    def span(self, i0, i1):
        """flat slice covering parameters i0 .. i1-1 (every one of them gets a view)."""
        for i in range(i0, i1):
            self.get(i)
        end = self.offsets[i1] if i1 < len(self.offsets) else self.flat.numel()
        return self.flat[self.offsets[i0]:end]

    def result(self):
        return tuple(self.views)

    # ---- one arena for all the autograd nodes of a backward pass that differentiate the same parameters --------------------------------------
    # BERT runs three to four times per step (ITM per retrieval sub-task, CAP) on the same parameters.  With an arena per node autograd sums
    # the nodes' gradients parameter by parameter (~900 add launches and 3 extra arena fills per step).  Inside ONE graph task the nodes
    # therefore share one arena: every node accumulates straight into it (the kernels add anyway), and a parameter's view is handed to
    # autograd by the FIRST node that touched it - later nodes return None for it (an undefined gradient is skipped by the engine, and the
    # parameter's AccumulateGrad only runs once all the nodes that feed it have, i.e. after the last in-place accumulation).  Parameters no
    # node touched keep grad None exactly as before.
    _shared = {}

    # The scheme needs the handed-out view to stay autograd's accumulator for the parameter until the last node has run: true as long as only
    # these nodes feed it.  A parameter with OTHER consumers (BERT's word embeddings are the LM head's tied decoder weight) is `private`: every
    # node gets a buffer of its own for it and autograd sums them as it always did (a second defined gradient makes the engine replace its
    # accumulator by an out-of-place sum, after which in-place additions to the view would be lost).
    @classmethod
    def session(cls, params, groups=None, private=()):
        gid = torch._C._current_graph_task_id() if hasattr(torch._C, "_current_graph_task_id") else -1
        if gid < 0 or not runtime.CFG.share_grad_arena:
            return _ArenaSession(cls(params, groups))
        for k in [k for k in cls._shared if k[0] != gid]:    # arenas of finished backward passes
            del cls._shared[k]
        key = (gid, id(params[0]), len(params))
        arena = cls._shared.get(key)
        if arena is None:
            arena = cls._shared[key] = cls(params, groups)
            arena.handed = set()
        return _ArenaSession(arena, private)


class _ArenaSession:
    """One autograd node's window on a (possibly shared) GradArena: same accessors; result() hands out the views of the parameters THIS node
    touched and nobody handed out before."""

    def __init__(self, arena, private=()):
        self.arena, self.touched = arena, set()
        self.flat = arena.flat
        self.private, self.own, self.own_fused = frozenset(private), {}, {}

    def get(self, i):
        if i in self.private:
            if i not in self.own:
                self.own[i] = torch.zeros(self.arena.shapes[i], dtype=torch.float32, device=self.arena.flat.device)
            return self.own[i]
        self.touched.add(i)
        return self.arena.get(i)

    def fused(self, idxs, shape):
        if any(i in self.private for i in idxs):   # a private group: ONE buffer of this node's own, the members' views adjacent in it
            assert all(i in self.private for i in idxs), "a fused group is private as a whole"
            key = tuple(idxs)
            if key not in self.own_fused:
                sizes = [math.prod(self.arena.shapes[i]) for i in idxs]
                buf = torch.zeros(sum(sizes), dtype=torch.float32, device=self.arena.flat.device)
                o = 0
                for i, n in zip(idxs, sizes):
                    assert i not in self.own
                    self.own[i] = buf[o:o + n].view(self.arena.shapes[i])
                    o += n
                self.own_fused[key] = buf
            return self.own_fused[key].view(shape)
        self.touched.update(idxs)
        return self.arena.fused(idxs, shape)

    def span(self, i0, i1):
        assert not any(i in self.private for i in range(i0, i1))
        self.touched.update(range(i0, i1))
        return self.arena.span(i0, i1)

    def result(self):
        handed = getattr(self.arena, "handed", None)
        if handed is None:
            return self.arena.result()
        out = tuple(self.own[i] if i in self.own else (self.arena.views[i] if (i in self.touched and i not in handed) else None)
                    for i in range(len(self.arena.views)))
        handed.update(self.touched)
        return out


def linear_wgrad(dy16, x16, dw, inv_s, n_out=None, n_in=None, dbias=None):
    """dw[N_out, N_in] += inv_s * dy16^T x16   (reduction over the rows);  dbias[N_out] += inv_s * column sums of dy16 in the same launch
    (the weight-gradient kernel stages the dy panel anyway; mico_gemm_epilogue::colsum_out)."""
    n_out = n_out or dw.shape[0]
    n_in = n_in or dw.shape[1]
    ops.gemm(dy16, x16, dw, ta=True, tb=True, M=n_out, N=n_in, K=dy16.shape[0], accumulate=True, alpha=inv_s,
             split_k=0, colsum_out=dbias)   # 0 = let the library size the K split in whole waves of resident workgroups


def _ln_fold_ok(a, dt):
    """The block's LayerNorm inputs are kept as fp16 normalised rows (activation diet level 3) and the step multiplies in fp16: a weight gradient
    against a LayerNorm OUTPUT can be taken against those rows directly (_wgrad_ln_folded)."""
    return bool(a.get("xn", False)) and dt == torch.float16 and runtime.CFG.ln_fold_wgrad


def _wgrad_ln_folded(dy16, xhat16, gamma, beta, dw, db, inv_s, dbt=None):
    """dw += inv_s * dy16^T (xhat16 * gamma + beta),  db += inv_s * colsum(dy16)  without forming the LayerNorm output:
    dy^T (xhat gamma + beta) = (dy^T xhat) . gamma[n] + colsum(dy)[m] beta[n]  (mico_dw_colfold).  The reference's autograd multiplies against the
    saved LayerNorm output (eva_vit_model.py:409-416 -> nn.Linear backward); here xhat is rounded to fp16 once and the affine map stays in fp32.
    dbt: a zeroed fp32 [n_out] buffer that receives this launch's bias gradient (then db is not touched)."""
    n_out, n_in = dw.shape
    dwt = torch.zeros((n_out, n_in), dtype=torch.float32, device=dy16.device)
    own = dbt is None
    if own:
        dbt = torch.zeros(n_out, dtype=torch.float32, device=dy16.device)
    linear_wgrad(dy16, xhat16, dwt, inv_s, dbias=dbt)
    ops.dw_colfold(dwt, dbt, gamma.detach(), beta.detach(), dw, db if own else None)


# ======================================================================================================================
# EVA ViT tower  (reference: model/evaclip/eva_vit_model.py:611-650 forward_features, :409-416 Block, :293-365 Attention,
# :190-224 Mlp / SwiGLU, :427-448 PatchEmbed)
# ======================================================================================================================
class TowerSpec:
    def __init__(self, arch, names, grid, rope=None):
        self.arch = arch
        self.names = names
        self.idx = {n: i for i, n in enumerate(names)}
        self.D = arch["width"]
        self.H = arch["heads"]
        self.hd = self.D // self.H
        self.P = arch["patch"]
        self.grid = grid
        self.np = grid * grid
        self.N = self.np + 1
        self.hidden = arch["mlp_hidden"]
        # GEMM operands need 16-byte rows: a hidden width that is not a multiple of 8 (EVA02-CLIP-L: int(1024 * 2.6667) = 2730) lives
        # in buffers padded to a multiple of 64 with zero weights / biases / LayerNorm parameters behind it - algebraically the
        # same MLP (the padded units compute silu(0) * 0 = 0 and feed zero weights)
        self.hidden_pad = self.hidden if self.hidden % 8 == 0 else (self.hidden + 63) // 64 * 64
        self.eps = 1e-6
        self.rope = rope   # (cos, sin) fp32 [np, hd] device tensors or None
        self.kpad3 = (3 * self.P * self.P + 63) // 64 * 64
        self.kpad1 = (self.P * self.P + 63) // 64 * 64
        self._ranges = None

    def block_param_ranges(self):
        """{block index: (first, last + 1) position in `names`} - a block's parameters are contiguous in registration order."""
        if self._ranges is None:
            r = {}
            for k, n in enumerate(self.names):
                if n.startswith("blocks."):
                    b = int(n.split(".")[1])
                    lo, hi = r.get(b, (k, k))
                    assert hi == k, "block parameters are expected to be contiguous"
                    r[b] = (lo, k + 1)
            self._ranges = r
        return self._ranges

    def non_block_param_ranges(self):
        br = sorted(self.block_param_ranges().values())
        out, prev = [], 0
        for lo, hi in br:
            if lo > prev:
                out.append((prev, lo))
            prev = hi
        if prev < len(self.names):
            out.append((prev, len(self.names)))
        return out


def _padv(p, n):
    """fp32 vector zero-padded to n entries (bias / LayerNorm parameter of a padded hidden width)"""
    v = p.detach()
    if v.shape[0] == n:
        return v
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.shape[0]] = v
    return out


class _Operand16:
    """A 16-bit GEMM operand buffer that carries its block-scaled fp8 copy (fp8 mode, produced by the LayerNorm that wrote it)."""
    __slots__ = ("t", "mx8")

    def __init__(self, t):
        self.t, self.mx8 = t, None


def _ln16(x, g, b, eps, rows, cols, dt, dev, frame_map=None, rows_per_frame=0, x_copy=None, valid_cols=0, mx8_for=None, xhat=None,
          x_normalized=False):
    """LayerNorm -> 16-bit GEMM operand.  Returns (buf, view, mean, rstd); in the split-precision (fp16 parity) mode buf is
    [rows, 2*cols] = [hi | lo] and view its hi half.  frame_map: compacting gather of whole frames (rows = kept rows).
    xhat: fp16 [rows, cols] buffer that receives the normalised rows (what the backward keeps instead of an fp32 copy, TowerDiet.xh16);
    x_normalized: x IS such a buffer - the output is re-created from it (no statistics: mean / rstd come back as None)."""
    split = runtime.split_activations() and cols % 64 == 0
    buf = _empty((rows, 2 * cols if split else cols), dt, dev)
    mean, rstd = (None, None) if x_normalized else (_empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev))
    if mx8_for is not None and not split and not valid_cols and runtime.fp8_enabled() and runtime.CFG.fp8_fused_quant and cols % 128 == 0 \
            and cols <= 2048 and _mx8_worthwhile(rows, mx8_for):
        # fp8 mode: the GEMM this LayerNorm feeds takes a block-scaled fp8 operand - written by the LayerNorm itself (== quant_mx8 of the
        # 16-bit output, without that pass); the operand rides on the buffer object to _gemm_fwd
        buf = _Operand16(buf)
        buf.mx8 = ops.layernorm_fwd_mx8(x, g, b, eps, out16=buf.t, mean=mean, rstd=rstd, dtype=dt, frame_map=frame_map,
                                        rows_per_frame=rows_per_frame, x_copy=x_copy, xhat16=xhat, x_normalized=x_normalized)
        return buf, buf.t, mean, rstd
    ops.layernorm_fwd(x, g, b, eps, out16=buf, mean=mean, rstd=rstd, split16=split, dtype=dt, frame_map=frame_map,
                      rows_per_frame=rows_per_frame, x_copy=x_copy, valid_cols=valid_cols, xhat16=xhat, x_normalized=x_normalized)
    return buf, (buf[:, :cols] if split else buf), mean, rstd


def _mx8_worthwhile(m, n):
    """The fp8 kernel has one tile shape (256x256, one workgroup per CU): it pays where the problem fills the chip - the towers' GEMMs
    and BERT's cross-attention K/V projections - not on BERT's few thousand text rows, which stay on the 128x128 16-bit kernel."""
    return ((m + 255) // 256) * ((n + 255) // 256) >= 128


def _gemm_fwd(a_buf, a_cols, plist, tag, out, k_pad=None, n_pad=None, ln=True, **epi):
    """Forward GEMM out = a W^T: plain in the bf16 configuration; in the fp16 parity configuration the weight is hi|lo split
    (and the activation too when its producer emitted [hi | lo]) and the products are summed by one k-segmented launch."""
    pre = None
    if isinstance(a_buf, _Operand16):
        pre, a_buf = a_buf.mx8, a_buf.t
    if runtime.fp8_enabled() and a_cols % 128 == 0 and "pos" not in epi and not k_pad and not n_pad and _mx8_worthwhile(a_buf.shape[0], out.shape[1]):
        # configs[4]: block-scaled fp8 MFMA.  The activation is quantised by the LayerNorm that produced it (round 4: _ln16(mx8_for=...)) or
        # here in a pass of its own (the GELU / attention outputs: 5 TB/s), the weight's fp8 copy is cached per optimizer step.
        return ops.gemm_mx8(pre if pre is not None else ops.quant_mx8(a_buf[:, :a_cols]), runtime.gemm_weight_mx8(plist, tag), out,
                            dtype=a_buf.dtype, **epi)
    w, ks = runtime.gemm_weight(plist, tag, k_pad=k_pad, n_pad=n_pad, ln_fed=ln)
    if ks is not None and a_buf.shape[1] == 2 * a_cols and a_cols == ks[0]:
        ks = (ks[0], [0, a_cols, 0], [0, 0, ks[0]])
    return ops.gemm(a_buf[:, :a_cols], w, out, ksegs=ks, **epi)


def _gemm_dx(dy16, plist, tag, out, k_pad=None, n_pad=None, ln=True, **epi):
    """Input gradient out = epilogue(dy W), W = the rows of `plist` concatenated ([N_out, K_in]): 16-bit MFMA reading W reduction-major
    (no transposed copy), or - configs[4] - the block-scaled fp8 MFMA on dy and a transposed fp8 copy of W, both quantised along N_out."""
    n_out = n_pad or sum(p.shape[0] for p in plist)          # (padded hidden widths: zero rows / columns behind the parameter's own)
    k_in = k_pad or plist[0].numel() // plist[0].shape[0]
    if k_pad or n_pad:
        return ops.gemm(dy16, runtime.gemm_weight(plist, tag, k_pad=k_pad, n_pad=n_pad)[0], out, tb=True, M=dy16.shape[0], N=k_in, K=n_out, **epi)
    if runtime.fp8_enabled() and n_out % 128 == 0 and _mx8_worthwhile(dy16.shape[0], k_in):
        return ops.gemm_mx8(ops.quant_mx8(dy16), runtime.gemm_weight_mx8(plist, tag, transposed=True), out, dtype=dy16.dtype, **epi)
    return ops.gemm(dy16, runtime.gemm_weight(plist, tag, ln_fed=ln)[0], out, tb=True, M=dy16.shape[0], N=k_in, K=n_out, **epi)


def _aux_buf(M, N, K, dt, dev):
    """(buffer, tiled) for an MLP's gelu'(pre-activation) tensor between its forward GEMM and its backward (ops.aux_buffer).  Row-major where one
    of the two launches leaves the 16-bit persistent kernel: fp8 mode (the dX launch is an MX-fp8 one), the parity configuration (activation
    [hi | lo]: three k-segments; the weights-split head blocks of the timed precision - x W_hi + x W_lo on the wrapped A stream - stay tiled).
    The same buffer carries the PRE-ACTIVATION where the block keeps that instead (_mlp_keeps_pre)."""
    return ops.aux_buffer(M, N, K, dt, dev, tiled_ok=not (runtime.fp8_enabled() or runtime.split_activations()))


def _mlp_keeps_pre():
    """A plain-MLP tower block that keeps its MLP intermediates keeps ONE 16-bit tensor - the fc1 pre-activation h - instead of gelu(h) and gelu'(h)
    (runtime.CFG.mlp_keep_pre, round 6): fc2's input-gradient launch multiplies by gelu'(h) computed from it and re-creates gelu(h) for fc2's weight
    gradient in the same epilogue (MICO_ACT_GELU_GRAD with aux_in and aux_out).  Not in fp8 mode (its dX launch has the multiply epilogue only)."""
    return runtime.CFG.mlp_keep_pre and not runtime.fp8_enabled()


def _qkv_params(P, b, arch):
    if arch["subln"]:
        return [P(b + "attn.q_proj.weight"), P(b + "attn.k_proj.weight"), P(b + "attn.v_proj.weight")]
    return [P(b + "attn.qkv.weight")]


class TowerDiet:
    """What one tower pass keeps per block for its hand-written backward (runtime.set_activation_diet, tower_plan).
    keep_mlp[i]: block i keeps its two MLP intermediates (GELU output, GELU': 4 hidden bytes per token) - otherwise the backward re-runs fc1
    with the GELU-pair epilogue;  keep_ln[i]: it keeps its two LayerNorm outputs (4 D) - otherwise they are re-created from the saved
    LayerNorm input rows;  xh16: those input rows are kept as fp16 NORMALISED rows (x - mean) * rstd (mico_ln_fwd_params::xhat16, 2 x 2 D
    bytes per token and block) instead of fp32 copies (2 x 4 D): all the LayerNorm backward needs next to rstd, and y = xhat gamma + beta
    re-creates the output.  The forward's values do not depend on any of this; with xh16 the backward sees the normalised rows rounded to fp16
    (relative 2^-12: its LayerNorm outputs can differ from the forward's in the last bit - far inside the gradient tolerances, but not the
    bit-for-bit recompute of the fp32 copies, which is why a step that fits without it does not use it).
    level: the legacy uniform levels 0 (keep all) / 1 (no MLP intermediates) / 2 (nor LayerNorm outputs) with fp32 rows, 3 = xh16 rows with
    nothing else kept except the MLP intermediates of the LAST mlp_blocks blocks (the first ones the backward frees)."""
    __slots__ = ("xh16", "keep_mlp", "keep_ln", "level", "mlp_blocks", "mlp_pre")

    def __init__(self, depth, level=0, mlp_blocks=0, keep=None, mlp_pre=None):
        """mlp_blocks: at level 3, the LAST n blocks keep their MLP intermediates; keep (level 3): an explicit set of block indices instead
        (tower_plan picks them by what their recompute costs per byte)."""
        level = int(level)
        assert 0 <= level <= 3
        self.level, self.xh16 = level, level == 3
        if level == 3 and keep is not None:
            kset = {int(i) for i in keep if 0 <= int(i) < depth}
        elif level == 3:
            kset = set(range(depth - min(depth, max(0, int(mlp_blocks))), depth))
        else:
            kset = set(range(depth)) if level == 0 else set()
        self.keep_mlp = [i in kset for i in range(depth)]
        self.mlp_blocks = len(kset)
        self.keep_ln = [level <= 1] * depth
        # what a kept block keeps of its MLP: the fc1 pre-activation alone (2 hidden bytes per token; fc2's dX launch re-creates gelu and gelu' from
        # it: ~0.4 ms per block dearer than the stored pair's launches) or gelu + gelu' (4 hidden bytes).  tower_plan takes the pair when every block
        # can keep it (configs[2], configs[4]: everything fits) and the pre-activation when memory is what limits the kept blocks (one rank of
        # configs[3]); a forced level 3 follows runtime.CFG.mlp_keep_pre.
        self.mlp_pre = (level == 3 and _mlp_keeps_pre()) if mlp_pre is None else (bool(mlp_pre) and _mlp_keeps_pre())

    @classmethod
    def of(cls, diet, depth):
        return diet if isinstance(diet, cls) else cls(depth, diet or 0)

    def describe(self):
        if self.level < 3:
            return self.level
        kept = [i for i, k in enumerate(self.keep_mlp) if k]
        what = "fc1 pre-activation" if self.mlp_pre else "MLP intermediates"
        return f"3 (fp16 normalised rows; {what} kept in {len(kept)} of {len(self.keep_mlp)} blocks: {_ranges(kept)})"


def _ranges(idx):
    """[0, 1, 2, 3, 31, 32, 39] -> '0-3, 31-32, 39'"""
    out, i = [], 0
    while i < len(idx):
        j = i
        while j + 1 < len(idx) and idx[j + 1] == idx[j] + 1:
            j += 1
        out.append(str(idx[i]) if i == j else f"{idx[i]}-{idx[j]}")
        i = j + 1
    return ", ".join(out) if out else "none"


def _h2d(t, dev):
    """Host tensor -> device without a stream synchronisation: torch's blocking copy from pageable memory ends in a hipStreamSynchronize - at the
    top of a step that drains everything the host had queued (tools/probes/sync_probe.py found exactly two such calls per step, both DropPlan's).
    The pinned staging buffer comes from torch's caching host allocator, which keeps it alive until the copy has run."""
    if torch.device(dev).type != "cuda":
        return t.to(dev)
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


class DropPlan:
    """Host-side plan of one step's stochastic depth (eva_vit_model.py:121-138 drop_path, applied per frame at :409-416).
    A dropped residual branch contributes exactly zero to the forward value and to every gradient, so the engine does not
    evaluate it: per (block, branch) the kept frames are compacted (LayerNorm gathers them, the output-projection epilogue
    scatters them back onto the fp32 residual stream in place) and every GEMM / attention launch in between runs on
    kept_frames * N rows.  scale [depth, 2, Bf] holds 0 or 1/keep; it is best drawn on the host (no device sync)."""

    skip_dropped = True      # False: evaluate every branch and multiply by the 0 / 1/keep scale (the reference's schedule)
    stats = [0, 0]           # running (kept, total) branch-frame counts, for bench.py's executed-FLOP accounting

    def __init__(self, scale, n_frames, dev):
        cpu = scale.detach().to("cpu", torch.float32)
        assert cpu.shape[1] == 2 and cpu.shape[2] == n_frames, (tuple(cpu.shape), n_frames)
        self.scale = _h2d(cpu, dev) if scale.device.type == "cpu" else scale.detach().to(dev, torch.float32).contiguous()
        keep = (cpu != 0).reshape(-1, n_frames)
        if not DropPlan.skip_dropped:
            keep = torch.ones_like(keep)
        self.counts = keep.sum(-1).tolist()
        DropPlan.stats[0] += sum(self.counts)
        DropPlan.stats[1] += keep.numel()
        frames = keep.nonzero()[:, 1].to(torch.int32)          # row-major: sorted by (block, branch), then frame
        self.frames = _h2d(frames, dev)
        self.starts = [0]
        for c in self.counts:
            self.starts.append(self.starts[-1] + c)
        self.n_frames = n_frames
        self._keep, self._dev, self._tr = keep, dev, None      # (the hand-over tables are built when the backward first asks for one)

    def _build_transitions(self):
        # Backward hand-over between consecutive branches (j = 2 block + which is processed in DEScending order): the LayerNorm backward of
        # branch j produces the residual-stream gradient that branch j - 1's GEMMs read as a compact 16-bit operand over ITS kept frames.
        # For every j >= 1: where each kept frame of j sits in j - 1's compact list (-1: not kept there), and the frames only j - 1 keeps.
        # Built lazily (no-grad forwards and the plan that only prices a chunked step never need them), vectorised over all branches, and
        # shipped in ONE host -> device copy.
        keep = self._keep
        nb = keep.shape[0]
        pos = keep.long().cumsum(1) - 1
        cur, prev, ppos = keep[1:], keep[:-1], pos[:-1]
        jj, fx = cur.nonzero(as_tuple=True)                                # kept frames of branch j = jj + 1, row-major
        dst = torch.where(prev[jj, fx], ppos[jj, fx], torch.full_like(fx, -1))
        oj, of = (prev & ~cur).nonzero(as_tuple=True)                      # frames only branch j - 1 keeps
        n_dst, n_diff = cur.sum(1).tolist(), (prev & ~cur).sum(1).tolist()
        self.tr_dst_starts, self.tr_diff_starts = [0], [0]
        for a_, d_ in zip(n_dst, n_diff):
            self.tr_dst_starts.append(self.tr_dst_starts[-1] + a_)
            self.tr_diff_starts.append(self.tr_diff_starts[-1] + d_)
        flat = _h2d(torch.cat((dst, of, ppos[oj, of])).to(torch.int32), self._dev)
        n1, n2 = dst.numel(), of.numel()
        self.tr_dst, self.tr_diff_frames, self.tr_diff_dst = flat[:n1], flat[n1:n1 + n2], flat[n1 + n2:]
        self._tr = nb

    def transition(self, block, which):
        """Hand-over from branch j = 2 block + which to branch j - 1 in the backward: (slot of each of j's kept frames in j - 1's compact
        list or -1, the frames only j - 1 keeps, their slots) as int32 device tensors; None for j = 0."""
        j = block * 2 + which
        if j == 0:
            return None
        if self._tr is None:
            self._build_transitions()
        a0, a1 = self.tr_dst_starts[j - 1], self.tr_dst_starts[j]
        d0, d1 = self.tr_diff_starts[j - 1], self.tr_diff_starts[j]
        return self.tr_dst[a0:a1], self.tr_diff_frames[d0:d1], self.tr_diff_dst[d0:d1]

    def branch(self, block, which):
        """-> (kept frame count, int32 device frame list or None when every frame is kept, [Bf] scale vector)"""
        j = block * 2 + which
        cnt = self.counts[j]
        fmap = None if cnt == self.n_frames else self.frames[self.starts[j]:self.starts[j] + cnt]
        return cnt, fmap, self.scale[block, which]

    def kept_fraction(self):
        return sum(self.counts) / float(len(self.counts) * self.n_frames)


_TAIL_EXP = None   # experiment hook (tools/precision_probe.py --tail)


# ----------------------------------------------------------------------------------------------------------------------
# The POST-norm block of EVA02-CLIP-bigE-14-plus (mico.py:341-344; eva_vit_model.py:411-413):
#     x = x + drop_path(norm1(attn(x)));  x = x + drop_path(norm2(mlp(x)))
# Off the ViT-g hot path - built from the hot path's kernels without its frame compaction, activation diet and
# gradient hand-over: the branch input is a 16-bit cast of the fp32 stream, the branch output stays fp32 for its LayerNorm, every branch is
# evaluated and scaled per frame (the reference's own drop_path schedule).
# ----------------------------------------------------------------------------------------------------------------------
def _postnorm_block_forward(spec, P, b, x, Bf, dps, a, dt, dev, strides3):
    D, N, H, hd, Hd = spec.D, spec.N, spec.H, spec.hd, spec.hidden
    M = Bf * N

    def branch_out(br, pre, sc):
        y = _empty((M, D), torch.float32, dev)
        mean, rstd = _empty((M,), torch.float32, dev), _empty((M,), torch.float32, dev)
        ops.layernorm_fwd(br, P(b + pre + ".weight"), P(b + pre + ".bias"), spec.eps, out32=y, mean=mean, rstd=rstd, dtype=dt)
        if sc is not None:
            y = y.view(Bf, N, D) * sc.view(Bf, 1, 1)
        return x + y.view(M, D), mean, rstd

    # attention branch
    x16a = _empty((M, D), dt, dev)
    ops.cast_f32_to_16(x, x16a)
    qb, vb = P(b + "attn.q_bias").detach(), P(b + "attn.v_bias").detach()
    qkv = _empty((M, 3 * D), dt, dev)
    _gemm_fwd(x16a, D, [P(b + "attn.qkv.weight")], "qkv", qkv, bias=torch.cat((qb, torch.zeros_like(qb), vb)), ln=False)
    ao = _empty((M, D), dt, dev)
    lse = _empty((Bf, H, N), torch.float32, dev)
    ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ao, lse, B=Bf, H=H, Sq=N, Sk=N, hd=hd, scale=hd ** -0.5, **strides3)
    br1 = _empty((M, D), torch.float32, dev)
    _gemm_fwd(ao, D, [P(b + "attn.proj.weight")], "w", br1, bias=P(b + "attn.proj.bias"), ln=False)
    sc1 = dps[0] if dps is not None else None
    x, mean1, rstd1 = branch_out(br1, "norm1", sc1)
    # MLP branch
    x16b = _empty((M, D), dt, dev)
    ops.cast_f32_to_16(x, x16b)
    act, h = _empty((M, Hd), dt, dev), _empty((M, Hd), dt, dev)
    _gemm_fwd(x16b, D, [P(b + "mlp.fc1.weight")], "w", act, bias=P(b + "mlp.fc1.bias"), aux_out=h, act=ops.ACT_GELU_SAVE_DERIV, ln=False)
    br2 = _empty((M, D), torch.float32, dev)
    _gemm_fwd(act, Hd, [P(b + "mlp.fc2.weight")], "w", br2, bias=P(b + "mlp.fc2.bias"), ln=False)
    sc2 = dps[1] if dps is not None else None
    x, mean2, rstd2 = branch_out(br2, "norm2", sc2)
    a.update(post=True, x16a=x16a, qkv=qkv, ao=ao, lse=lse, br1=br1, mean1=mean1, rstd1=rstd1, sc1=sc1,
             x16b=x16b, act=act, h=h, br2=br2, mean2=mean2, rstd2=rstd2, sc2=sc2)
    return x


def _postnorm_block_backward(spec, P, G, b, a, g, Bf, dt, dev, strides3, S):
    """g: fp32 gradient of the residual stream [M, D] at the block's output, updated in place to the gradient at its input."""
    D, N, H, hd, Hd = spec.D, spec.N, spec.H, spec.hd, spec.hidden
    M = Bf * N
    inv_s = 1.0 / S

    def branch_in(br, mean, rstd, pre, sc):
        """d(branch output), 16-bit in the gradient scale S: the LayerNorm backward of sc * g"""
        dy = g if sc is None else (g.view(Bf, N, D) * sc.view(Bf, 1, 1)).view(M, D)
        d16 = _empty((M, D), dt, dev)
        ops.layernorm_bwd(dy, br, P(b + pre + ".weight"), mean, rstd, dx16=d16, scale16=S, dgamma=G(b + pre + ".weight"),
                          dbeta=G(b + pre + ".bias"), dtype=dt)
        return d16

    # MLP branch
    d16 = branch_in(a["br2"], a["mean2"], a["rstd2"], "norm2", a["sc2"])
    w1, w2 = P(b + "mlp.fc1.weight"), P(b + "mlp.fc2.weight")
    linear_wgrad(d16, a["act"], G(b + "mlp.fc2.weight"), inv_s, dbias=G(b + "mlp.fc2.bias"))
    dh = a["act"]
    _gemm_dx(d16, [w2], "w", dh, ln=False, aux_in=a["h"], act=ops.ACT_MUL_AUX)
    linear_wgrad(dh, a["x16b"], G(b + "mlp.fc1.weight"), inv_s, dbias=G(b + "mlp.fc1.bias"))
    _gemm_dx(dh, [w1], "w", g, ln=False, alpha=inv_s, resid=g)
    del d16, dh
    # attention branch
    d16 = branch_in(a["br1"], a["mean1"], a["rstd1"], "norm1", a["sc1"])
    wp = P(b + "attn.proj.weight")
    linear_wgrad(d16, a["ao"], G(b + "attn.proj.weight"), inv_s, dbias=G(b + "attn.proj.bias"))
    dao = _empty((M, D), dt, dev)
    _gemm_dx(d16, [wp], "w", dao, ln=False)
    qkv = a["qkv"]
    dqkv = _empty((M, 3 * D), dt, dev)
    delta = _empty((Bf, H, N), torch.float32, dev)
    ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], a["ao"], dao, a["lse"], dqkv, dqkv[:, D:], dqkv[:, 2 * D:], delta,
                 B=Bf, H=H, Sq=N, Sk=N, hd=hd, scale=hd ** -0.5, **strides3)
    dbias = torch.zeros(3 * D, dtype=torch.float32, device=dev)
    linear_wgrad(dqkv, a["x16a"], G(b + "attn.qkv.weight"), inv_s, dbias=dbias)
    G(b + "attn.q_bias").add_(dbias[:D])
    G(b + "attn.v_bias").add_(dbias[2 * D:])
    _gemm_dx(dqkv, [P(b + "attn.qkv.weight")], "qkv", g, ln=False, alpha=inv_s, resid=g)


def _tower_forward(spec, groups, dp_scale, params, save, diet=0, plan=None):
    """One pass of the tower over the frames in `groups`.  save=False keeps nothing for a backward (chunked forward).
    diet (plain-MLP towers): 1 = the MLP intermediates are not kept (the backward recomputes fc1 + GELU / GELU'), 2 = nor the LayerNorm
    outputs (recomputed from the saved fp32 rows) - see runtime.set_activation_diet."""
    dt = runtime.compute_dtype()
    P = lambda n: params[spec.idx[n]]
    dev = params[0].device
    D, N, np_, H, hd = spec.D, spec.N, spec.np, spec.H, spec.hd
    arch = spec.arch
    depth = arch["depth_built"]
    Bf = sum(g.shape[0] for g in groups)
    M = Bf * N
    if arch.get("postnorm"):
        plan, diet = None, 0          # (the post-norm tower evaluates every branch and scales it: no frame compaction, no diet)
        if dp_scale is not None:
            dp_scale = dp_scale.detach().to(dev, torch.float32)
    elif plan is None:
        plan = DropPlan(dp_scale, Bf, dev) if dp_scale is not None else None
    if arch["swiglu"]:
        diet = 0      # (the SwiGLU towers keep everything: B/16 and L/14 frames are an order of magnitude smaller)
    diet = TowerDiet.of(diet, depth)
    xh16 = diet.xh16 and save
    x = _empty((M, D), torch.float32, dev)
    # ---- patch embedding: im2row + GEMM(+bias +pos, patch rows -> token rows) ; CLS rows ----
    pe_w, pe_b, pos = P("patch_embed.proj.weight"), P("patch_embed.proj.bias"), P("pos_embed")
    pos2 = pos.detach().reshape(N, D)
    saved_rows = []
    f0 = 0
    for g in groups:
        C = g.shape[1]
        kpad = spec.kpad3 if C == 3 else spec.kpad1
        w16, ks = runtime.gemm_weight([pe_w], "pe3" if C == 3 else "pe1", k_pad=kpad, channel_sum=(C != 3))
        rows16 = _empty((g.shape[0] * np_, kpad), dt, dev)
        ops.im2row(g.contiguous().float(), rows16, spec.P, kpad)
        ops.gemm(rows16, w16, x[f0 * N:], M=g.shape[0] * np_, N=D, K=kpad, bias=pe_b, pos=pos2, pos_rows=N,
                 remap=(np_, 1, 1), ksegs=ks)
        if save:
            saved_rows.append(rows16)
        f0 += g.shape[0]
    ops.cls_rows(x, Bf, N, P("cls_token").detach().reshape(D), pos2[0])

    def branch_io(i, which):
        """-> (kept frames, frame list | None, scale vector | None)"""
        return plan.branch(i, which) if plan is not None else (Bf, None, None)

    def strides3():
        return dict(q_strides=(N * 3 * D, 3 * D), k_strides=(N * 3 * D, 3 * D), v_strides=(N * 3 * D, 3 * D), o_strides=(N * D, D))

    acts = []
    Hd = spec.hidden
    pstate = runtime.snapshot()   # per-block precision: runtime.enter_block; the callers restore the pass-wide state
    for i in range(depth):
        b = f"blocks.{i}."
        a = {}
        runtime.enter_block(i, pstate, bool(arch["swiglu"]), depth)
        if _TAIL_EXP is not None:
            runtime.CFG.split_fp16 = (i >= depth - _TAIL_EXP[0]) if _TAIL_EXP[0] >= 0 else (i < -_TAIL_EXP[0])
            runtime.CFG.split_mode = _TAIL_EXP[1]
        if arch.get("postnorm"):
            x = _postnorm_block_forward(spec, P, b, x, Bf, dp_scale[i] if dp_scale is not None else None, a, dt, dev, strides3())
            if save:
                acts.append(a)
            del a
            continue
        # --- attention branch: x <- x + s1 * proj(attn(LN1 x)) on the kept frames ---
        B1, fmap1, sc1 = branch_io(i, 0)
        a.update(B1=B1, fmap1=fmap1, sc1=sc1, tr1=plan.transition(i, 0) if plan is not None else None)
        if B1 > 0:
            M1 = B1 * N
            xc1 = _empty((M1, D), torch.float32, dev) if (fmap1 is not None and save and not xh16) else None
            xh1 = _empty((M1, D), torch.float16, dev) if xh16 else None
            ln1b, ln1, mean1, rstd1 = _ln16(x, P(b + "norm1.weight"), P(b + "norm1.bias"), spec.eps, M1, D, dt, dev,
                                            frame_map=fmap1, rows_per_frame=N, x_copy=xc1, xhat=xh1, mx8_for=3 * D)
            qb, vb = P(b + "attn.q_bias").detach(), P(b + "attn.v_bias").detach()
            qkv_bias = torch.cat((qb, torch.zeros_like(qb), vb))
            qkv = _empty((M1, 3 * D), dt, dev)
            _gemm_fwd(ln1b, D, _qkv_params(P, b, arch), "qkv", qkv, bias=qkv_bias)
            if spec.rope is not None:
                ops.rope(qkv, N * 3 * D, 3 * D, B1, N, H, hd, spec.rope[0], spec.rope[1])
                ops.rope(qkv[:, D:], N * 3 * D, 3 * D, B1, N, H, hd, spec.rope[0], spec.rope[1])
            ao = _empty((M1, D), dt, dev)
            lse = _empty((B1, H, N), torch.float32, dev)
            ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ao, lse, B=B1, H=H, Sq=N, Sk=N, hd=hd, scale=hd ** -0.5, **strides3())
            proj_in = ao
            if arch["subln"]:
                proj_in, aln, mean_a, rstd_a = _ln16(ao, P(b + "attn.inner_attn_ln.weight"), P(b + "attn.inner_attn_ln.bias"),
                                                     spec.eps, M1, D, dt, dev)
                a.update(aln=aln, mean_a=mean_a, rstd_a=rstd_a)
            # every frame kept: out of place, the input buffer itself is the saved LN input; otherwise the epilogue
            # scatters the kept frames onto the stream in place and the LN's compact copy (xc1) is what is saved
            # (with the fp16 normalised rows kept instead, nothing reads the stream again: in place either way)
            x_mid = _empty((M, D), torch.float32, dev) if (fmap1 is None and save and not xh16) else x
            _gemm_fwd(proj_in, D, [P(b + "attn.proj.weight")], "w", x_mid, bias=P(b + "attn.proj.bias"), resid=x, row_scale=sc1,
                      rows_per_scale=N, row_map=fmap1, rows_per_map=N, ln=bool(arch["subln"]))
            a.update(x1=xh1 if xh16 else (x if fmap1 is None else xc1), xn=xh16, mean1=mean1, rstd1=rstd1, ln1=ln1 if diet.keep_ln[i] else None,
                     qkv=qkv, ao=ao, lse=lse)
            x = x_mid
        # --- MLP branch ---
        B2, fmap2, sc2 = branch_io(i, 1)
        a.update(B2=B2, fmap2=fmap2, sc2=sc2, tr2=plan.transition(i, 1) if plan is not None else None)
        if B2 > 0:
            M2 = B2 * N
            xc2 = _empty((M2, D), torch.float32, dev) if (fmap2 is not None and save and not xh16) else None
            xh2 = _empty((M2, D), torch.float16, dev) if xh16 else None
            ln2b, ln2, mean2, rstd2 = _ln16(x, P(b + "norm2.weight"), P(b + "norm2.bias"), spec.eps, M2, D, dt, dev,
                                            frame_map=fmap2, rows_per_frame=N, x_copy=xc2, xhat=xh2, mx8_for=None if arch["swiglu"] else spec.hidden)
            x_out = _empty((M, D), torch.float32, dev) if (fmap2 is None and save and not xh16) else x
            epi = dict(resid=x, row_scale=sc2, rows_per_scale=N, row_map=fmap2, rows_per_map=N)
            if arch["swiglu"]:
                Hp = spec.hidden_pad                     # == Hd unless the hidden width needs padding (see TowerSpec)
                npad = Hp if Hp != Hd else None
                b1, b2 = _padv(P(b + "mlp.w1.bias"), Hp), _padv(P(b + "mlp.w2.bias"), Hp)
                g1, g2 = _empty((M2, Hp), dt, dev), _empty((M2, Hp), dt, dev)
                if runtime.split_precision():
                    # parity configuration: the gate runs in fp32 (x1, x2 and the gated product never round to fp16 on the
                    # forward path); 16-bit copies of x1 / x2 are kept for the backward kernels only
                    x1f, x2f = _empty((M2, Hp), torch.float32, dev), _empty((M2, Hp), torch.float32, dev)
                    _gemm_fwd(ln2b, D, [P(b + "mlp.w1.weight")], "w", x1f, bias=b1, n_pad=npad)
                    _gemm_fwd(ln2b, D, [P(b + "mlp.w2.weight")], "w", x2f, bias=b2, n_pad=npad)
                    hsw = _empty((M2, Hp), torch.float32, dev)
                    ops.swiglu_fwd_f32(x1f, x2f, hsw)
                    ops.cast_f32_to_16(x1f, g1)
                    ops.cast_f32_to_16(x2f, g2)
                    del x1f, x2f
                else:
                    _gemm_fwd(ln2b, D, [P(b + "mlp.w1.weight")], "w", g1, bias=b1, n_pad=npad)
                    _gemm_fwd(ln2b, D, [P(b + "mlp.w2.weight")], "w", g2, bias=b2, n_pad=npad)
                    hsw = _empty((M2, Hp), dt, dev)
                    ops.swiglu_fwd(g1, g2, hsw)
                hlnb, hln, mean_f, rstd_f = _ln16(hsw, _padv(P(b + "mlp.ffn_ln.weight"), Hp), _padv(P(b + "mlp.ffn_ln.bias"), Hp), spec.eps,
                                                  M2, Hp, dt, dev, valid_cols=Hd if npad else 0)
                _gemm_fwd(hlnb, Hp, [P(b + "mlp.w3.weight")], "w", x_out, bias=P(b + "mlp.w3.bias"), k_pad=npad, **epi)
                a.update(g1=g1, g2=g2, hsw=hsw, hln=hln, mean_f=mean_f, rstd_f=rstd_f)
            else:
                act = _empty((M2, Hd), dt, dev)
                if save and diet.keep_mlp[i] and diet.mlp_pre:
                    # the block keeps its pre-activation (2 hidden bytes per token): GELU with the pre-activation copy leaving from the registers
                    h, h_tiled = _aux_buf(M2, Hd, D, dt, dev)
                    _gemm_fwd(ln2b, D, [P(b + "mlp.fc1.weight")], "w", act, bias=P(b + "mlp.fc1.bias"), aux_out=h, act=ops.ACT_GELU, aux_tiled=h_tiled)
                    a.update(h=h, h_tiled=h_tiled, pre=True)
                    del h
                elif (save and diet.keep_mlp[i]) or runtime.fp8_enabled() or not runtime.CFG.fc1_plain_gelu:   # (the 8-phase fp8 kernel has the pair epilogue only)
                    h, h_tiled = _aux_buf(M2, Hd, D, dt, dev)
                    _gemm_fwd(ln2b, D, [P(b + "mlp.fc1.weight")], "w", act, bias=P(b + "mlp.fc1.bias"), aux_out=h, act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=h_tiled)
                    if diet.keep_mlp[i]:
                        a.update(h=h, act=act, h_tiled=h_tiled)
                    del h
                else:   # nobody will read GELU' (no backward, or the activation diet recomputes the pair): half the output bytes of the launch
                    _gemm_fwd(ln2b, D, [P(b + "mlp.fc1.weight")], "w", act, bias=P(b + "mlp.fc1.bias"), act=ops.ACT_GELU)
                _gemm_fwd(act, Hd, [P(b + "mlp.fc2.weight")], "w", x_out, bias=P(b + "mlp.fc2.bias"), ln=False, **epi)
                del act
            a.update(x2=xh2 if xh16 else (x if fmap2 is None else xc2), xn=xh16, mean2=mean2, rstd2=rstd2, ln2=ln2 if diet.keep_ln[i] else None,
                     ln2b=ln2b if (diet.keep_ln[i] and not diet.keep_mlp[i]) else None)
            x = x_out
        if save:
            acts.append(a)
        del a
    runtime.restore(pstate)
    out = _empty((M, D), torch.float32, dev)
    mean_n, rstd_n = _empty((M,), torch.float32, dev), _empty((M,), torch.float32, dev)
    ops.layernorm_fwd(x, P("norm.weight"), P("norm.bias"), spec.eps, out32=out, mean=mean_n, rstd=rstd_n, dtype=dt)
    saved = None
    if save:
        saved = dict(acts=acts, dt=dt, final=(x, mean_n, rstd_n), groups_meta=[(g.shape[0], g.shape[1]) for g in groups],
                     saved_rows=saved_rows, Bf=Bf)
    return out.view(Bf, N, D), saved


def _tower_backward(spec, params, saved, dout, grads, final=True):
    """Backward of one _tower_forward(save=True) pass; parameter gradients are accumulated into the shared `grads` arena.
    final: no later pass adds to these gradients (last chunk) - finished blocks are then announced to runtime.grad_slice_hook."""
    hook = runtime.grad_slice_hook() if final else None
    block_range = spec.block_param_ranges()
    dt = saved["dt"]
    P = lambda n: params[spec.idx[n]]
    dev = dout.device
    D, N, np_, H, hd, Bf = spec.D, spec.N, spec.np, spec.H, spec.hd, saved["Bf"]
    arch = spec.arch
    M = Bf * N
    S = runtime.grad_scale()
    inv_s = 1.0 / S

    def G(name, like=None):
        return grads.get(spec.idx[name])

    def w16_of(p):
        return runtime.gemm_weight([p])[0]

    strides3 = dict(q_strides=(N * 3 * D, 3 * D), k_strides=(N * 3 * D, 3 * D), v_strides=(N * 3 * D, 3 * D),
                    o_strides=(N * D, D))
    x_last, mean_n, rstd_n = saved["final"]
    g = _empty((M, D), torch.float32, dev)   # running gradient of the fp32 residual stream (updated in place)
    ops.layernorm_bwd(dout.contiguous().view(M, D), x_last, P("norm.weight"), mean_n, rstd_n, dx32=g,
                      dgamma=G("norm.weight"), dbeta=G("norm.bias"), dtype=dt)
    saved["final"] = None
    del x_last
    Hd = spec.hidden
    pstate = runtime.snapshot()
    fuse = runtime.CFG.fuse_grad_handover

    def take_g16(pre, rows, fmap, sc):
        """The branch's 16-bit gradient operand [rows, D] = S * sc[frame] * g over its kept frames: handed over by the previous LayerNorm
        backward (only the frames that branch did not keep are still gathered), or gathered whole."""
        if pre is not None:
            g16, dframes, ddst = pre
            if dframes is not None and dframes.numel() > 0:
                ops.gather_rows_cast(g, g16, row_scale=sc, rows_per_scale=N, scale=S, frame_map=dframes, rows_per_frame=N, dst_map=ddst)
            return g16
        g16 = _empty((rows, D), dt, dev)
        ops.gather_rows_cast(g, g16, row_scale=sc, rows_per_scale=N, scale=S, frame_map=fmap, rows_per_frame=N)
        return g16

    def handover(tr, nb, nsc, had_plan):
        """LayerNorm-backward arguments that make it write the next branch's operand (nb kept frames, per-frame scale nsc), and the `pre`
        tuple for take_g16().  tr: DropPlan.transition of the current branch (None without a plan: both branches keep every frame)."""
        if not fuse or nb <= 0 or (had_plan and tr is None):
            return {}, None
        g16n = _empty((nb * N, D), dt, dev)
        if tr is None:
            return dict(dx16=g16n, scale16=S), (g16n, None, None)
        return dict(dx16=g16n, scale16=S, dx16_dst=tr[0], dx16_frame_scale=nsc), (g16n, tr[1], tr[2])

    pre = None
    for i in reversed(range(arch["depth_built"])):
        b = f"blocks.{i}."
        a = saved["acts"].pop()
        runtime.enter_block(i, pstate, bool(arch["swiglu"]), arch["depth_built"])   # the block's weights in the layout its forward used
        if a.get("post"):
            _postnorm_block_backward(spec, P, G, b, a, g, Bf, dt, dev, strides3, S)
            del a
            if hook is not None:
                i0, i1 = block_range[i]
                hook(grads.span(i0, i1), params[i0:i1])
            continue
        # ---------------- MLP branch (kept frames only: a dropped branch has no gradient) ----------------
        if a["B2"] > 0:
            M2, fmap2 = a["B2"] * N, a["fmap2"]
            g16 = take_g16(pre, M2, fmap2, a["sc2"])
            pre = None
            # gradient at the LayerNorm output: 16-bit like every other gradient operand (dqkv, dH, g16 carry the same scale) - the input-gradient
            # GEMM writes half the bytes and the LayerNorm backward reads half; the SwiGLU path accumulates two GEMMs into it and stays fp32
            dln2 = _empty((M2, D), torch.float32 if (arch["swiglu"] or not runtime.CFG.ln_grad_16bit) else dt, dev)
            if arch["swiglu"]:
                w1, w2, w3 = P(b + "mlp.w1.weight"), P(b + "mlp.w2.weight"), P(b + "mlp.w3.weight")
                Hp = spec.hidden_pad
                npad = Hp if Hp != Hd else None
                pend = []

                def GP(name, shape):
                    """gradient target of `name`, or - padded hidden width - a zero-filled buffer of the padded shape whose leading
                    block is added to the parameter's gradient afterwards"""
                    tgt = G(name)
                    if npad is None:
                        return tgt
                    tmp = torch.zeros(shape, dtype=torch.float32, device=dev)
                    pend.append((tgt, tmp))
                    return tmp

                linear_wgrad(g16, a["hln"], GP(b + "mlp.w3.weight", (D, Hp)), inv_s)
                ops.colsum(g16, G(b + "mlp.w3.bias"), scale=inv_s, accumulate=True)
                dhln = _empty((M2, Hp), dt, dev)
                _gemm_dx(g16, [w3], "w", dhln, k_pad=npad)
                dhsw = _empty((M2, Hp), dt, dev)
                ops.layernorm_bwd(dhln, a["hsw"], _padv(P(b + "mlp.ffn_ln.weight"), Hp), a["mean_f"], a["rstd_f"], dx16=dhsw,
                                  dgamma=GP(b + "mlp.ffn_ln.weight", (Hp,)), dbeta=GP(b + "mlp.ffn_ln.bias", (Hp,)), grad_scale=inv_s,
                                  dtype=dt, valid_cols=Hd if npad else 0)
                dx1, dx2 = dhln, _empty((M2, Hp), dt, dev)   # reuse dhln storage for dx1
                ops.swiglu_bwd(a["g1"], a["g2"], dhsw, dx1, dx2)
                linear_wgrad(dx1, a["ln2"], GP(b + "mlp.w1.weight", (Hp, D)), inv_s)
                linear_wgrad(dx2, a["ln2"], GP(b + "mlp.w2.weight", (Hp, D)), inv_s)
                ops.colsum(dx1, GP(b + "mlp.w1.bias", (Hp,)), scale=inv_s, accumulate=True)
                ops.colsum(dx2, GP(b + "mlp.w2.bias", (Hp,)), scale=inv_s, accumulate=True)
                _gemm_dx(dx1, [w1], "w", dln2, n_pad=npad)
                _gemm_dx(dx2, [w2], "w", dln2, n_pad=npad, accumulate=True)
                for tgt, tmp in pend:
                    tgt.add_(tmp[tuple(slice(0, n) for n in tgt.shape)])
                del dhln, dhsw, dx1, dx2, pend
            else:
                w1, w2 = P(b + "mlp.fc1.weight"), P(b + "mlp.fc2.weight")
                fold2 = False
                kept_pre = bool(a.get("pre"))      # the block kept its fc1 pre-activation only (_mlp_keeps_pre)
                if "act" not in a and not kept_pre:
                    # activation diet: the LayerNorm output (from the saved rows - fp32: compact kept rows or the whole stream, in either case
                    # exactly the M2 rows the forward normalised; or their fp16 normalised form), then fc1 + GELU / GELU' as the forward ran them
                    ln2b = a["ln2b"]
                    if ln2b is None:
                        ln2b, a["ln2"], _, _ = _ln16(a["x2"], P(b + "norm2.weight"), P(b + "norm2.bias"), spec.eps, M2, D, dt, dev, mx8_for=Hd,
                                                     x_normalized=a["xn"])
                    (a["h"], a["h_tiled"]), a["act"] = _aux_buf(M2, Hd, D, dt, dev), _empty((M2, Hd), dt, dev)
                    _gemm_fwd(ln2b, D, [w1], "w", a["act"], bias=P(b + "mlp.fc1.bias"), aux_out=a["h"], act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=a["h_tiled"])
                    del ln2b
                elif a["ln2"] is None:      # the MLP intermediates were kept, the LayerNorm output (fc1's weight gradient reads it) was not
                    fold2 = _ln_fold_ok(a, dt)   # ... and need not be: the gradient is taken against the normalised rows (below)
                    if not fold2:
                        _, a["ln2"], _, _ = _ln16(a["x2"], P(b + "norm2.weight"), P(b + "norm2.bias"), spec.eps, M2, D, dt, dev, x_normalized=a["xn"])
                if kept_pre:
                    # dH = (g16 W2) . gelu'(h) with gelu(h) re-created by the same epilogue - then fc2's weight gradient against it
                    dh, gact = _empty((M2, Hd), dt, dev), _empty((M2, Hd), dt, dev)
                    _gemm_dx(g16, [w2], "w", dh, ln=False, aux_in=a["h"], aux_out=gact, act=ops.ACT_GELU_GRAD, aux_tiled=a["h_tiled"])
                    a["h"] = None
                    linear_wgrad(g16, gact, G(b + "mlp.fc2.weight"), inv_s, dbias=G(b + "mlp.fc2.bias"))
                    del gact
                else:
                    linear_wgrad(g16, a["act"], G(b + "mlp.fc2.weight"), inv_s, dbias=G(b + "mlp.fc2.bias"))
                    dh = a["act"]   # the GELU output is dead after the weight gradient: reuse its storage for dH
                    _gemm_dx(g16, [w2], "w", dh, ln=False, aux_in=a["h"], act=ops.ACT_MUL_AUX, aux_tiled=a["h_tiled"])   # a["h"] = gelu'(pre-activation)
                if fold2:
                    _wgrad_ln_folded(dh, a["x2"], P(b + "norm2.weight"), P(b + "norm2.bias"), G(b + "mlp.fc1.weight"), G(b + "mlp.fc1.bias"), inv_s)
                else:
                    linear_wgrad(dh, a["ln2"], G(b + "mlp.fc1.weight"), inv_s, dbias=G(b + "mlp.fc1.bias"))
                _gemm_dx(dh, [w1], "w", dln2)
                del dh
            ho, pre = handover(a["tr2"], a["B1"], a["sc1"], a["sc2"] is not None)
            ops.layernorm_bwd(dln2, a["x2"], P(b + "norm2.weight"), a["mean2"], a["rstd2"], dy_scale=inv_s, dx_add=g,
                              dx32=g, dgamma=G(b + "norm2.weight"), dbeta=G(b + "norm2.bias"), dtype=dt, frame_map=fmap2,
                              rows_per_frame=N, x_normalized=a.get("xn", False), **ho)
            del dln2, g16
        else:
            pre = None
        # ---------------- attention branch ----------------
        if a["B1"] > 0:
            B1, fmap1 = a["B1"], a["fmap1"]
            M1 = B1 * N
            g16 = take_g16(pre, M1, fmap1, a["sc1"])
            pre = None
            wp = P(b + "attn.proj.weight")
            proj_in = a["aln"] if arch["subln"] else a["ao"]
            linear_wgrad(g16, proj_in, G(b + "attn.proj.weight"), inv_s, dbias=G(b + "attn.proj.bias"))
            dao = _empty((M1, D), dt, dev)
            _gemm_dx(g16, [wp], "w", dao, ln=bool(arch["subln"]))
            if arch["subln"]:
                dao2 = _empty((M1, D), dt, dev)
                ops.layernorm_bwd(dao, a["ao"], P(b + "attn.inner_attn_ln.weight"), a["mean_a"], a["rstd_a"], dx16=dao2,
                                  dgamma=G(b + "attn.inner_attn_ln.weight"), dbeta=G(b + "attn.inner_attn_ln.bias"),
                                  grad_scale=inv_s, dtype=dt)
                dao = dao2
            fold1 = a["ln1"] is None and not arch["subln"] and _ln_fold_ok(a, dt)
            if a["ln1"] is None and not fold1:      # activation diet: the LayerNorm output was not kept
                _, a["ln1"], _, _ = _ln16(a["x1"], P(b + "norm1.weight"), P(b + "norm1.bias"), spec.eps, M1, D, dt, dev, x_normalized=a["xn"])
            qkv = a["qkv"]
            dqkv = _empty((M1, 3 * D), dt, dev)
            delta = _empty((B1, H, N), torch.float32, dev)
            ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], a["ao"], dao, a["lse"], dqkv, dqkv[:, D:], dqkv[:, 2 * D:], delta,
                         B=B1, H=H, Sq=N, Sk=N, hd=hd, scale=hd ** -0.5, **strides3)
            if spec.rope is not None:
                ops.rope(dqkv, N * 3 * D, 3 * D, B1, N, H, hd, spec.rope[0], spec.rope[1], inverse=True)
                ops.rope(dqkv[:, D:], N * 3 * D, 3 * D, B1, N, H, hd, spec.rope[0], spec.rope[1], inverse=True)
            dbias = torch.zeros(3 * D, dtype=torch.float32, device=dev)      # filled by the qkv weight-gradient launch below
            if arch["subln"]:
                dwf = torch.zeros((3 * D, D), dtype=torch.float32, device=dev)
                linear_wgrad(dqkv, a["ln1"], dwf, inv_s, dbias=dbias)
                G(b + "attn.q_proj.weight").add_(dwf[:D])
                G(b + "attn.k_proj.weight").add_(dwf[D:2 * D])
                G(b + "attn.v_proj.weight").add_(dwf[2 * D:])
            elif fold1:     # against the kept normalised rows; norm1's gamma / beta applied to the [3 D, D] result
                _wgrad_ln_folded(dqkv, a["x1"], P(b + "norm1.weight"), P(b + "norm1.bias"), G(b + "attn.qkv.weight"), None, inv_s, dbt=dbias)
            else:
                linear_wgrad(dqkv, a["ln1"], G(b + "attn.qkv.weight"), inv_s, dbias=dbias)
            G(b + "attn.q_bias").add_(dbias[:D])
            G(b + "attn.v_bias").add_(dbias[2 * D:])
            dln1 = _empty((M1, D), dt if runtime.CFG.ln_grad_16bit else torch.float32, dev)
            _gemm_dx(dqkv, _qkv_params(P, b, arch), "qkv", dln1)
            nxt = saved["acts"][-1] if saved["acts"] else None     # block i - 1: its MLP branch is next
            ho, pre = handover(a["tr1"], nxt["B2"], nxt["sc2"], a["sc1"] is not None) if nxt is not None else ({}, None)
            ops.layernorm_bwd(dln1, a["x1"], P(b + "norm1.weight"), a["mean1"], a["rstd1"], dy_scale=inv_s, dx_add=g,
                              dx32=g, dgamma=G(b + "norm1.weight"), dbeta=G(b + "norm1.bias"), dtype=dt, frame_map=fmap1,
                              rows_per_frame=N, x_normalized=a.get("xn", False), **ho)
            del dqkv, dao, dln1, g16
        else:
            pre = None
        del a
        if hook is not None:      # every gradient of block i is final: its arena slice can be reduced now
            i0, i1 = block_range[i]
            hook(grads.span(i0, i1), params[i0:i1])
    runtime.restore(pstate)
    # ---------------- patch embedding ----------------
    dpos = torch.zeros(N * D, dtype=torch.float32, device=dev)
    ops.colsum(g, dpos, rows=Bf, cols=N * D, ld=N * D)
    G("pos_embed").add_(dpos.view(1, N, D))
    G("cls_token").add_(dpos[:D].view(1, 1, D))
    pe_w = P("patch_embed.proj.weight")
    PP = spec.P * spec.P
    f0 = 0
    for (nf, C), rows16 in zip(saved["groups_meta"], saved["saved_rows"]):
        gp16 = _empty((nf * np_, D), dt, dev)
        ops.gather_rows_cast(g[f0 * N:], gp16, remap=(np_, 1, 1), scale=S)
        kpad = rows16.shape[1]
        dw = torch.zeros((D, kpad), dtype=torch.float32, device=dev)
        linear_wgrad(gp16, rows16, dw, inv_s)
        if C == 3:
            G("patch_embed.proj.weight").add_(dw[:, :3 * PP].reshape(D, 3, spec.P, spec.P))
        else:
            G("patch_embed.proj.weight").add_(dw[:, :PP].reshape(D, 1, spec.P, spec.P))
        ops.colsum(gp16, G("patch_embed.proj.bias"), scale=inv_s, accumulate=True)
        f0 += nf
    saved["saved_rows"] = None
    if hook is not None:              # what is left: embeddings, cls / pos, final norm (before and after the block parameters)
        for i0, i1 in spec.non_block_param_ranges():
            hook(grads.span(i0, i1), params[i0:i1])


def _slice_groups(groups, c0, c1):
    """The frames [c0, c1) of the concatenated modality groups, as a list of per-group slices."""
    out, f0 = [], 0
    for g in groups:
        f1 = f0 + g.shape[0]
        lo, hi = max(c0, f0), min(c1, f1)
        if lo < hi:
            out.append(g[lo - f0:hi - f0])
        f0 = f1
    return out


_SOFT_FRAC = float(os.environ.get("MICO_HBM_SOFT_FRAC", "0.87"))
# Level 3 fills what its smaller rows free with MLP intermediates - up to the soft budget minus this margin: the budget's activation estimate is
# exact for them (4 hidden bytes per kept MLP token) while the levels' own bytes are priced with slack (kept + 0.02), so a plan filled to the
# brim peaks ~10 GiB above the same budget's level-2 plan (measured on one rank of configs[3]: 237 against 227 GiB with 14 blocks kept)
_MLP_KEEP_MARGIN = int(float(os.environ.get("MICO_MLP_KEEP_MARGIN_GIB", "6")) * (1 << 30))


def tower_plan(spec, n_frames, device, kept=1.0, block_tokens=None):
    """-> (frames per tower pass, TowerDiet).  Saved activations of the pre-norm plain-MLP tower cost, per kept token and block: two LayerNorm
    input copies (fp32: 8 D bytes; as fp16 normalised rows, TowerDiet.xh16: 4 D), the two LayerNorm outputs 4 D, qkv 6 D, the attention output
    2 D, the two MLP intermediates 4 hidden - depth * N * (20 D + 4 hidden) = 542 MB per frame for ViT-g/14 when everything is kept.  When the
    step's frames do not fit in the memory budget there are two ways to pay with recomputation, priced in tower-forward units:
      * the activation diet - level 1 drops the MLP intermediates (the backward re-runs fc1 + GELU: +0.33 of a tower forward), level 2 the
        LayerNorm outputs as well (two more LayerNorm passes per block: +0.02); level 3 (round 5) additionally keeps the LayerNorm input rows as
        fp16 normalised rows (no recomputation at all: 4 D fewer bytes than level 2) and spends what that frees on the MLP intermediates of as
        many blocks as fit - the LAST ones, which the backward frees first (each saves its share of the 0.33);
      * chunks - the tower runs in n equal chunks, forward without saving except for the last one, and in the backward the other chunks are
        recomputed with saving and differentiated: +(n - 1) / n.
    The cheapest combination that fits is taken; levels 0-2 recompute the forward's own values bit for bit and win ties.  One rank of BASELINE
    configs[3] (14 frames per sample, 896 per GPU at b = 64, ~0.8 kept by stochastic depth) runs in ONE pass at level 3 with the MLP
    intermediates of ~10 of the 40 blocks kept (round 3: level 2, 166 GB of activations; round 2: three chunks, +0.67).
    kept: fraction of the (block, branch, frame) triples the step's stochastic-depth draw keeps (only those are evaluated and saved);
    block_tokens (optional, per block): kept MLP-branch tokens of each block (DropPlan counts x N) - prices the per-block MLP choice exactly.
    The post-norm tower (bigE) and the SwiGLU towers keep everything (no diet): only chunks are priced for them, at their own bytes per frame."""
    if torch.device(device).type != "cuda":
        from ._lib import MicoHipError
        raise MicoHipError("the ViT tower runs on an MI355X device only (parameters are on %s): mico_amd has no CPU path" % device)
    depth, N, D, Hd = spec.arch["depth_built"], spec.N, spec.D, spec.hidden
    postnorm = bool(spec.arch.get("postnorm"))
    dietable = not spec.arch["swiglu"] and not postnorm
    # (the post-norm block saves fp32 br1 / br2 (8 D), x16a / x16b / ao (6 D), qkv (6 D), act + h (4 hidden): the same 20 D + 4 hidden per token)
    pre_ok = dietable and _mlp_keeps_pre()      # a kept block may keep its fc1 pre-activation alone (2 bytes per hidden unit) instead of gelu + gelu' (4)
    per_frame = {0: depth * N * (20 * D + 4 * Hd)}
    if dietable:
        per_frame[1] = depth * N * 20 * D
        per_frame[2] = depth * N * 16 * D
        per_frame[3] = depth * N * 12 * D
    extra = {0: 0.0, 1: 0.33, 2: 0.35, 3: 0.35}
    forced_chunk, forced_diet = runtime.tower_chunk_override(), runtime.activation_diet_override()
    forced_level = forced_diet[0] if forced_diet is not None else None
    levels = [forced_level if forced_level in per_frame else 0] if forced_diet is not None else sorted(per_frame)
    if forced_chunk:
        lv = levels[0] if forced_diet is not None else 0
        return forced_chunk, TowerDiet(depth, lv, forced_diet[1] if (forced_diet is not None and lv == 3) else 0)      # (forced level 3: CFG.mlp_keep_pre decides the stash)
    free, _ = torch.cuda.mem_get_info(device)
    free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)    # cached blocks are reusable
    # what the rest of the step needs next to the tower's saved activations - the backward's temporaries, BERT with its cross-attention K/V
    # over frames x N condition tokens, the gradient arena: measured 23 GiB at 320 frames (configs[2]: 162 GiB peak, 139 of it activations)
    # and 57 GiB at 896 (one rank of configs[3]: 231 GiB peak at level 2, 266 GiB at level 1)
    # A staged step (MiCo.forward(backward_scale=...), round 6) builds and differentiates the BERT passes one condition set at a time: one K/V memory,
    # one set of graphs and gradient buffers on top of the tower stash instead of all of them - measured on one rank of configs[3]: 36.8 GiB between
    # the end of encode_batch and the step's peak (inside the va set's backward) against 56 for the direct form.
    headroom = (12 << 30) + n_frames * ((30 << 20) if runtime.step_staged else (52 << 20))
    # Two budgets for the saved activations.  HARD: what fits at all - 0.90 of the free memory beyond the headroom (0.95 put one rank of
    # configs[3] on level 1 at a 268 of 288 GiB peak).  SOFT: what keeps the step's projected peak (allocated now + headroom + activations)
    # under MICO_HBM_SOFT_FRAC of the device memory - the margin a data-parallel job needs for RCCL's buffers, a second reducer and allocator
    # fragmentation (level 1 "fitted" one rank of configs[3] at 266-268 GiB ALLOCATED; level 2 costs 0.02 of a tower forward
    # more and peaks at 231).  0.82 until late in round 5, 0.87 since: with level 3 every 0.01 is one more block's MLP intermediates kept,
    # and one rank of configs[3] measured 40.56 / 40.66 samples/s at 0.82 (12 blocks, 250 GiB RESERVED by the caching allocator), 41.28 / 41.00
    # at 0.86 (15, 262 GiB), 41.47 / 41.47 at 0.90 (18, 274 GiB), no allocator retry in any; 24 forced blocks ran out of memory.  The soft budget decides unless staying under it would need chunked recomputation
    # that the hard budget avoids (> 0.05 of a tower forward dearer): memory margin is worth a LayerNorm recompute, not half a forward.
    total = torch.cuda.get_device_properties(device).total_memory
    hard = int(0.90 * max(free - headroom, free // 4))
    soft_frac = _SOFT_FRAC
    if "MICO_HBM_SOFT_FRAC" not in os.environ and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # a rank of an N > 1 job: RCCL's buffers, the reducer's fp32 buckets and the fragmentation they cause have only been measured on a one-rank
        # group (no multi-GPU box this far): 0.03 of the device stay free for them until a scaling run confirms the one-rank figure
        soft_frac -= 0.03
    soft = min(hard, int(soft_frac * total) - torch.cuda.memory_allocated(device) - headroom)
    exact = block_tokens is not None and len(block_tokens) == depth      # the step's own stochastic-depth draw, counted: no slack needed
    if not exact:
        kept = min(1.0, max(0.05, kept + 0.02))      # (a little slack: the draw differs from chunk to chunk)
        block_tokens = [kept * n_frames * N] * depth
    # what re-running fc1 costs per kept token of block i, relative to a plain block: the head-split blocks of the timed precision
    # (runtime.CFG.head_split_blocks: x W_hi + x W_lo, K doubled) cost twice - their intermediates are the most valuable bytes to keep
    fc1_w = [runtime.fc1_recompute_weight(i, bool(spec.arch["swiglu"])) for i in range(depth)]
    mlp_work_total = float(sum(w * t for w, t in zip(fc1_w, block_tokens))) or 1.0

    def cheapest(budget):
        best = None
        for lv in levels:
            pf = per_frame[lv] * kept
            most = max(1, int(max(budget, 0) // pf))
            n_chunks = -(-n_frames // most)
            cost = extra[lv] + (n_chunks - 1) / n_chunks
            keep, pre = None, False
            if lv == 3 and forced_diet is not None:
                pre = pre_ok
                keep = list(range(depth - min(depth, max(0, forced_diet[1])), depth))
            elif lv == 3 and n_chunks == 1:
                # one pass at level 3: what is left of the budget keeps MLP intermediates, from the LAST block backwards - the backward starts
                # there, so kept buffers are released early and every later recompute reuses what the finished blocks freed.  (Keeping the
                # head-split blocks first - their fc1 recompute is twice as long, the most valuable bytes on paper - was measured: blocks 0-3 +
                # 34-39 against 28-39 at the same 230 GiB peak, 39.96 / 39.74 against 40.31 / 40.38 samples/s in alternating runs: buffers that
                # live until the END of the backward keep the allocator at the peak for the whole backward.)  A kept block saves its share of
                # the fc1 recompute, weighted by what its recompute costs.
                # (a staged step priced with the draw's own counts: the estimate was within 1 GiB of the measured peak at 39 and at 40 kept blocks - half
                # the margin)
                left0 = budget - pf * n_frames - (_MLP_KEEP_MARGIN // 2 if (runtime.step_staged and exact) else _MLP_KEEP_MARGIN)

                def fill(mlp_b):
                    # (round 6, with ONE tensor per kept block nearly every block fits: the blocks whose recompute is dearest first - the
                    # head-split blocks' fc1 has twice the reduction length - then from the last block backwards; a block that does not fit is
                    # skipped, not the end.  The pair keeps round 5's measured order: from the last block backwards, stop at the first misfit.)
                    order = sorted(range(depth), key=lambda i: (-fc1_w[i], -i)) if mlp_b == 2 else list(reversed(range(depth)))
                    left, kp, sv = left0, [], 0.0
                    for i in order:
                        need = block_tokens[i] * mlp_b * Hd
                        if need > left:
                            if mlp_b == 2:
                                continue
                            break
                        left -= need
                        sv += fc1_w[i] * block_tokens[i]
                        kp.append(i)
                    return kp, sv
                keep, saved = fill(4)
                # the pre-activation stash where memory limits the pair: its two launches per kept block (GELU + copy forward, GELU' + gelu(h) in
                # fc2's dX) cost ~0.04 of a block's forward more than the pair's - far less than the fc1 launch every additional kept block saves
                if pre_ok and len(keep) < depth:
                    keep2, saved2 = fill(2)
                    if saved2 > saved:
                        keep, saved, pre = keep2, saved2, True
                cost = 0.02 + 0.33 * (1.0 - saved / mlp_work_total) + (0.01 * len(keep) / depth if pre else 0.0)
            if best is None or cost < best[0] - 1e-9:
                best = (cost, -(-n_frames // n_chunks), lv, keep, pre)      # equal chunks: the one whose activations are kept is then as large as the others
        return best

    bs, bh = cheapest(soft), cheapest(hard)
    if bs[1] >= n_frames or bs[0] <= bh[0] + 0.05:
        best = bs       # (one pass under the soft budget: how many blocks keep their MLP intermediates is the soft budget's decision too)
    else:
        # only chunked recomputation keeps the step under the soft budget and the hard one avoids it: take the hard budget's level and chunking,
        # but none of its optional extras (MLP intermediates kept at level 3)
        best = (bh[0], bh[1], bh[2], bh[3] if (forced_diet is not None and bh[2] == 3) else None, bh[4])
    return best[1], TowerDiet(depth, best[2], keep=best[3] if best[2] == 3 else None, mlp_pre=best[4] if best[2] == 3 else False)


def tower_chunk_frames(spec, n_frames, device):
    """Frames per tower pass (see tower_plan)."""
    return tower_plan(spec, n_frames, device)[0]


class EvaTowerFn(torch.autograd.Function):
    _logged_plan = None

    @staticmethod
    def forward(ctx, spec, groups, dp_scale, *params):
        runtime.remember_precision(ctx)
        with runtime.using(runtime.snapshot()):   # the block loop switches the state per block (runtime.enter_block)
            return EvaTowerFn._forward(ctx, spec, groups, dp_scale, *params)

    @staticmethod
    def _forward(ctx, spec, groups, dp_scale, *params):
        Bf = sum(g.shape[0] for g in groups)
        needs_grad = any(ctx.needs_input_grad)    # False under torch.no_grad(): nothing is kept for a backward then
        depth = spec.arch["depth_built"]
        plan, chunk, diet = None, Bf, TowerDiet(depth, 0)
        if needs_grad:
            plan = DropPlan(dp_scale, Bf, params[0].device) if dp_scale is not None else None
            kept = plan.kept_fraction() if plan is not None else 1.0
            btok = [plan.counts[2 * i + 1] * spec.N for i in range(depth)] if (plan is not None and len(plan.counts) == 2 * depth) else None
            chunk, diet = tower_plan(spec, Bf, params[0].device, kept, btok)

            def record(**more):
                runtime.last_tower_plan = dict(frames=Bf, frames_per_pass=min(chunk, Bf), diet=diet.level, mlp_blocks_kept=diet.mlp_blocks,
                                               mlp_blocks=_ranges([i for i, k in enumerate(diet.keep_mlp) if k]) if diet.level == 3 else None,
                                               rows_fp16_normalised=diet.xh16, kept_fraction=kept,
                                               mlp_stash=("pre-activation (2 B per hidden unit)" if diet.mlp_pre else "gelu + gelu' (4 B per hidden unit)"), **more)
            record()
            if (Bf, min(chunk, Bf), diet.level, tuple(diet.keep_mlp)) != EvaTowerFn._logged_plan:   # once per distinct plan and process (= rank)
                EvaTowerFn._logged_plan = (Bf, min(chunk, Bf), diet.level, tuple(diet.keep_mlp))
                _log.info("tower plan: %d frames, %d per pass, activation diet %s (kept fraction %.3f)", Bf, min(chunk, Bf), diet.describe(), kept)
        ctx.spec, ctx.params, ctx.diet = spec, params, diet
        if chunk >= Bf:
            retry = False
            try:
                out, ctx.saved = _tower_forward(spec, groups, dp_scale, params, save=needs_grad, diet=diet, plan=plan)
                ctx.chunked = None
                return out
            except torch.cuda.OutOfMemoryError:
                # The plan is priced from a fitted headroom (tower_plan): another step shape (more condition tokens, other modality mixes) or a
                # smaller device can under-estimate it.  The forward only wrote buffers of its own, so it is repeated ONCE on the next more
                # conservative plan instead of failing the step.  Only the DECISION is taken here: while this handler is active the
                # exception's traceback keeps the failed _tower_forward frame - its `acts`, `x`, every buffer of the failed attempt - alive,
                # so empty_cache() would free nothing and the retry would allocate on top of them; the retry runs below, after
                # the handler has exited and the traceback is gone.
                if not needs_grad:
                    raise
                retry = True
            if retry:
                ctx.saved = None
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                dietable = not spec.arch["swiglu"] and not spec.arch.get("postnorm")
                if dietable and runtime.activation_diet_override() is None and (diet.level < 3 or diet.mlp_blocks > 0):
                    # next more conservative diet: no MLP intermediates at level 3, else the next level
                    diet = TowerDiet(depth, 3) if diet.level == 3 else TowerDiet(depth, diet.level + 1)
                else:
                    chunk = -(-Bf // 2)
                ctx.diet = diet
                record(oom_retry=True)
                _log.warning("tower pass ran out of memory: retrying with %d frames per pass, activation diet %s (rank-local decision)",
                             min(chunk, Bf), diet.describe())
                if chunk >= Bf:
                    out, ctx.saved = _tower_forward(spec, groups, dp_scale, params, save=needs_grad, diet=diet, plan=plan)
                    ctx.chunked = None
                    return out
        if plan is not None:      # the chunks draw their own plans from their slices of dp_scale: this one only priced the step
            DropPlan.stats[0] -= sum(plan.counts)
            DropPlan.stats[1] -= len(plan.counts) * Bf
        outs = []
        ctx.saved = None
        starts = list(range(0, Bf, chunk))
        for c0 in starts:
            c1 = min(Bf, c0 + chunk)
            sub_dp = dp_scale[:, :, c0:c1].contiguous() if dp_scale is not None else None
            keep = c0 == starts[-1]      # the last chunk's activations fit by construction: keep them, the backward starts there
            o, saved = _tower_forward(spec, _slice_groups(groups, c0, c1), sub_dp, params, save=keep, diet=diet)
            if keep:
                ctx.saved = saved
            outs.append(o)
        ctx.chunked = (groups, dp_scale, chunk, Bf)
        return torch.cat(outs, dim=0)

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dout):
        spec, params = ctx.spec, ctx.params
        runtime.mem_trace("tower backward start")
        grads = GradArena(params)
        if ctx.chunked is None:
            _tower_backward(spec, params, ctx.saved, dout, grads)
            ctx.saved = None
            runtime.mem_trace("tower backward end")
        else:
            groups, dp_scale, chunk, Bf = ctx.chunked
            starts = list(range(0, Bf, chunk))
            for c0 in reversed(starts):      # last chunk first: its activations were kept by the forward, the others are recomputed
                c1 = min(Bf, c0 + chunk)
                if c0 == starts[-1]:
                    saved, ctx.saved = ctx.saved, None
                else:
                    sub_dp = dp_scale[:, :, c0:c1].contiguous() if dp_scale is not None else None
                    _, saved = _tower_forward(spec, _slice_groups(groups, c0, c1), sub_dp, params, save=True, diet=ctx.diet)
                _tower_backward(spec, params, saved, dout[c0:c1], grads, final=(c0 == starts[0]))
                del saved
        return (None, None, None) + grads.result()


# ======================================================================================================================
# small fp32 functions (heads, pooling, normalisation) - exact fp32 kernels
# ======================================================================================================================
class _LinearF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        y = _empty((x2.shape[0], w.shape[0]), torch.float32, x.device)
        ops.sgemm(x2, w.detach(), y, bias=b.detach() if b is not None else None)
        ctx.save_for_backward(x2, w)
        ctx.has_b = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0]).contiguous().float()
        dx = _empty(x2.shape, torch.float32, dy.device)
        ops.sgemm(dy2, w.detach(), dx, tb=True)                  # dx = dy W
        dw = _empty(w.shape, torch.float32, dy.device)
        ops.sgemm(dy2, x2, dw, ta=True, tb=True)                 # dW = dy^T x
        db = None
        if ctx.has_b:
            db = torch.zeros(w.shape[0], dtype=torch.float32, device=dy.device)
            ops.colsum(dy2, db)
        return dx.view(ctx.xshape), dw, db


def linear_f32(x, w, b=None):
    return _LinearF32.apply(x, w, b)


class _MatmulNT(torch.autograd.Function):
    """a [m,k] @ b[n,k]^T * alpha (fp32, exact) - similarity matrices (vast.py:405-408)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = a.contiguous(), b.contiguous()
        y = _empty((a.shape[0], b.shape[0]), torch.float32, a.device)
        ops.sgemm(a, b, y, alpha=alpha)
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        da = db = None
        if ctx.needs_input_grad[0]:
            da = _empty(a.shape, torch.float32, dy.device)
            ops.sgemm(dy, b, da, tb=True, alpha=ctx.alpha)
        if ctx.needs_input_grad[1]:
            db = _empty(b.shape, torch.float32, dy.device)
            ops.sgemm(dy, a, db, ta=True, tb=True, alpha=ctx.alpha)
        return da, db, None


def matmul_nt(a, b, alpha=1.0):
    return _MatmulNT.apply(a, b, alpha)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        y = torch.empty_like(x)
        inv = _empty((x.shape[0],), torch.float32, x.device)
        ops.l2norm_fwd(x, y, inv)
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dx = torch.empty_like(y)
        ops.l2norm_bwd(dy.contiguous(), y, inv, dx)
        return dx


def l2_normalize(x):
    """F.normalize(x, dim=-1) (vast.py:245,253 ...)."""
    return _L2Norm.apply(x)


class _GeluF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous().float()
        y = torch.empty_like(x)
        ops.gelu_f32(x, y)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        ops.gelu_bwd_f32(x, dy.contiguous(), dx)
        return dx


def gelu_f32(x):
    return _GeluF32.apply(x)


class _LayerNormF32(torch.autograd.Function):
    """fp32 in / fp32 out LayerNorm for small tensors (Match_head, mico.py:49)."""

    @staticmethod
    def forward(ctx, x, g, b, eps):
        runtime.remember_precision(ctx)
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean, rstd = _empty((rows,), torch.float32, x.device), _empty((rows,), torch.float32, x.device)
        ops.layernorm_fwd(x2, g.detach(), b.detach(), eps, out32=y, mean=mean, rstd=rstd, dtype=runtime.compute_dtype())
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.xshape = x.shape
        return y.view(x.shape)

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dy):
        x2, g, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(x2.shape).contiguous().float()
        dx = torch.empty_like(x2)
        dg, db = torch.zeros_like(g, dtype=torch.float32), torch.zeros_like(g, dtype=torch.float32)
        ops.layernorm_bwd(dy2, x2, g.detach(), mean, rstd, dx32=dx, dgamma=dg, dbeta=db, dtype=runtime.compute_dtype())
        return dx.view(ctx.xshape), dg, db, None


def layer_norm_f32(x, g, b, eps):
    return _LayerNormF32.apply(x, g, b, eps)


class _ClsPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens):   # [b, n, N, D] fp32
        b, n, N, D = tokens.shape
        tokens = tokens.contiguous()
        out = _empty((b, D), torch.float32, tokens.device)
        ops.cls_pool_fwd(tokens, out, b, n, N * D, D)
        ctx.shape = (b, n, N, D)
        return out

    @staticmethod
    def backward(ctx, dy):
        b, n, N, D = ctx.shape
        dt = torch.zeros(ctx.shape, dtype=torch.float32, device=dy.device)
        ops.cls_pool_bwd(dy.contiguous(), dt, b, n, N * D, D)
        return dt


def cls_pool(tokens):
    """feature[:, :, 0].mean(1)  (mico.py:157-182)."""
    return _ClsPool.apply(tokens)


class _PoolVideo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens):   # [b, n, N, D]; the kernels are fp32-only and take raw pointers: any other dtype is converted here
        b, n, N, D = tokens.shape
        if not tokens.is_cuda:
            from ._lib import MicoHipError
            raise MicoHipError("pool_video runs on an MI355X device only (tokens are on %s): mico_amd has no CPU path" % tokens.device)
        ctx.in_dtype = tokens.dtype
        tokens = tokens.float().contiguous()
        out = _empty((b, n, 2, D), torch.float32, tokens.device)
        ops.pool_video_fwd(tokens, out, b * n, N, D)
        ctx.shape = (b, n, N, D)
        return out.to(ctx.in_dtype)

    @staticmethod
    def backward(ctx, dy):
        b, n, N, D = ctx.shape
        dt = _empty(ctx.shape, torch.float32, dy.device)
        ops.pool_video_bwd(dy.float().contiguous(), dt, b * n, N, D)
        return dt.to(ctx.in_dtype)


def pool_video(tokens):
    """torch.cat([x[:, :, 0:1], x[:, :, 1:].mean(2, keepdim=True)], dim=2)  (mico.py:190-191, 217-218, 233-234): per frame the CLS token and the
    mean of the patch tokens, one kernel each way."""
    return _PoolVideo.apply(tokens)


def mean_pool(tokens):
    """feature.mean(2).mean(1) - the Swin branch of pool_*_for_contra (mico.py:161-163): every frame has the same token count, so the mean of
    the per-frame means is the mean over all n * N token rows, which the CLS-pool kernel computes when each token is its own 'frame'."""
    b, n, N, D = tokens.shape
    return _ClsPool.apply(tokens.reshape(b, n * N, 1, D))


# ======================================================================================================================
# condition packing: Linear(Dv -> 768) + LN(1e-12) + frame embedding + type embedding   (mico.py:187-243, 400-403)
# ======================================================================================================================
class _CondPack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, w, b, g, beta, table, rows_per_group):
        """feats [rows, Dv] fp32; table [groups, 768] fp32 (frame + type embedding per frame slot);
        output row r gets table[(r // rows_per_group) % groups]."""
        runtime.remember_precision(ctx)
        dt = runtime.compute_dtype()
        dev = feats.device
        rows, Dv = feats.shape
        Dm = w.shape[0]
        x16 = _empty((rows, Dv), dt, dev)
        ops.cast_f32_to_16(feats.contiguous(), x16)
        w16, ks = runtime.gemm_weight([w])
        u = _empty((rows, Dm), torch.float32, dev)
        ops.gemm(x16, w16, u, bias=b.detach(), ksegs=ks)
        y = _empty((rows, Dm), torch.float32, dev)
        mean, rstd = _empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev)
        ops.layernorm_fwd(u, g.detach(), beta.detach(), 1e-12, out32=y, mean=mean, rstd=rstd, post_add=table.detach().contiguous(),
                          post_rows_per_group=rows_per_group, post_groups=table.shape[0], dtype=dt)
        ctx.save_for_backward(x16, u, mean, rstd, w, g)
        ctx.meta = (rows_per_group, table.shape[0], dt)
        return y

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dy):
        x16, u, mean, rstd, w, g = ctx.saved_tensors
        rpg, groups, dt = ctx.meta
        dev = dy.device
        rows, Dm = u.shape
        S = runtime.grad_scale()
        inv_s = 1.0 / S
        dy = dy.contiguous()
        du16 = _empty((rows, Dm), dt, dev)
        dg, dbeta = torch.zeros_like(g, dtype=torch.float32), torch.zeros_like(g, dtype=torch.float32)
        ops.layernorm_bwd(dy, u, g.detach(), mean, rstd, dx16=du16, scale16=S, dgamma=dg, dbeta=dbeta, dtype=dt)
        dw = torch.zeros(w.shape, dtype=torch.float32, device=dev)
        linear_wgrad(du16, x16, dw, inv_s)
        db = torch.zeros(Dm, dtype=torch.float32, device=dev)
        ops.colsum(du16, db, scale=inv_s)
        w16 = runtime.gemm_weight([w])[0]
        dx = _empty((rows, x16.shape[1]), torch.float32, dev)
        ops.gemm(du16, w16, dx, tb=True, M=rows, N=x16.shape[1], K=Dm, alpha=inv_s)
        # table gradient: rows are ordered (sample, frame slot, token): sum over samples, then over the tokens of a slot
        per = rpg * groups
        nb = rows // per
        tmp = torch.zeros(per * Dm, dtype=torch.float32, device=dev)
        ops.colsum(dy, tmp, rows=nb, cols=per * Dm, ld=per * Dm)
        dtable = torch.zeros((groups, Dm), dtype=torch.float32, device=dev)
        for f in range(groups):
            ops.colsum(tmp[f * rpg * Dm:], dtable[f], rows=rpg, cols=Dm, ld=Dm)
        return dx, dw, db, dg, dbeta, dtable, None


def cond_pack(feats, w, b, g, beta, table, rows_per_group):
    return _CondPack.apply(feats, w, b, g, beta, table, rows_per_group)


# ======================================================================================================================
# BERT with cross-attention (reference: model/bert.py:785-916 BertModel.forward, :393-461 BertLayer, :184-297 attention,
# :349-375 FFN, :101-149 embeddings).  Dropout is not applied (eval semantics).
# ======================================================================================================================
class BertSpec:
    def __init__(self, names, n_layers, heads=12, hidden=768, inter=3072, eps=1e-12):
        self.names = names
        self.idx = {n: i for i, n in enumerate(names)}
        self.L, self.H, self.D, self.I, self.eps = n_layers, heads, hidden, inter, eps
        # gradient views that the fused projections' weight-gradient GEMMs write as one matrix (GradArena groups)
        self.grad_groups = []
        for li in range(n_layers):
            for att, parts in (("attention", ("query", "key", "value")), ("crossattention", ("key", "value"))):
                for kind in ("weight", "bias"):
                    ns = [f"encoder.layer.{li}.{att}.self.{q}.{kind}" for q in parts]
                    if all(n in self.idx for n in ns):
                        self.grad_groups.append([self.idx[n] for n in ns])


def _fused_w(key, plist, dt_unused=None):
    return runtime.gemm_weight(plist, key)[0]


def _fwd_gemm(x, key, plist, out, **kw):
    kin = plist[0].shape[1]
    return _gemm_fwd(x, kin, plist, key, out, **kw)


# dropout sites of one BERT pass (the `site` of mico_dropout's counter hash): layer * 8 + {self-output, cross-output, FFN output,
# self-attention probabilities, cross-attention probabilities}; the embedding dropout has its own id
SITE_SELF_OUT, SITE_CROSS_OUT, SITE_FFN_OUT, SITE_SELF_P, SITE_CROSS_P, SITE_EMB = 0, 1, 2, 3, 4, 100000


def _kv_all_weights(kvparams, L):
    """key.weight, value.weight of every layer in order: the rows of the [L * 2 D, D] weight behind the interleaved K/V memory."""
    return [kvparams[4 * li + j] for li in range(L) for j in (0, 2)]


def _kv_layers(t2d, L, D):
    """[rows, L * 2 D] row-major shared K/V memory (or its gradient) -> its per-layer view [L][rows][2 D] (row stride L * 2 D)."""
    return t2d.view(t2d.shape[0], L, 2 * D).permute(1, 0, 2)


class DkvSession:
    """One per shared cross-attention K/V memory (BertModel.project_cross_kv attaches it to kv_own): the BERT passes of a step that read the same
    own set - the ITM triplet and the captioning pass - put its 16-bit gradient into ONE [n E, L * 2 D] buffer.  The first BertFn.backward of
    a backward pass writes the buffer, registers it here and hands it to autograd; the later ones add to it inside the short-query attention
    backward (mico_attn_params.dkv_accumulate) and return None for kv_own, so the engine holds exactly one gradient for CrossKVFn's output and
    never sums two 7 GB tensors out of place at the step's memory peak.  CrossKVFn.backward ends the session.
    closed: a pass could not take part (no short-query kernel for its shape) and returned a buffer of its own - the engine's sum replaces the
    registered buffer, so nothing may be added to it any more in this backward pass."""
    __slots__ = ("own", "closed")
    accumulated = 0      # BertFn backward passes that added to another pass's buffer (tests)

    def __init__(self):
        self.own, self.closed = None, False

    def reset(self):
        self.own, self.closed = None, False


class CrossKVFn(torch.autograd.Function):
    """Cross-attention K/V projections of the condition tokens for ALL layers, computed once per step and shared by every BERT pass
    that attends to the same tokens with the same weights: the ITM triplet [own | hard negative | own] (vast.py:438-447) contains
    the batch's own condition tokens twice and the captioning pass (vast.py:486-512) a third time - the reference projects them in
    every pass (bert.py:206-215).  Returns (kv_own, kv_neg): 16-bit views of ONE buffer whose rows are [own | neg], so that a BERT pass over the
    triplet reads it through mico_attn_params.kv_batch_mod.  Memory layout (runtime.CFG.kv_interleaved, default): [rows][L][K | V], handed out as
    2-D [n E, L * 2 D] tensors - a layer is a column block with row stride L * 2 D, every layer's projection one GEMM, and the backward one product
    over K = L * 2 D for the condition-token gradient; or [L][rows][K | V] (3-D [L][n E, 2 D] views) with a launch per layer (fp32 accumulation
    into the token gradient 12 times).
    The gradients handed back by BertFn for these two outputs stay in the engine's 16-bit gradient scale (runtime.grad_scale());
    this function removes it - the one place where a scaled gradient crosses autograd, between two of our own functions."""

    concat_backwards = 0      # backward passes that took the all-layers-at-once path of the interleaved layout (tests)

    @staticmethod
    def forward(ctx, spec, session, cond_own, cond_neg, *kvparams):
        runtime.remember_precision(ctx)
        ctx.session = session      # DkvSession or None: ended by the backward
        # kvparams: per layer key.weight, key.bias, value.weight, value.bias
        dt = runtime.compute_dtype()
        dev = cond_own.device
        n, E, D = cond_own.shape
        sets = 2 if cond_neg is not None else 1
        cond16 = _empty((sets * n * E, D), dt, dev)
        ops.cast_f32_to_16(cond_own.contiguous().view(n * E, D), cond16[:n * E])
        if cond_neg is not None:
            ops.cast_f32_to_16(cond_neg.contiguous().view(n * E, D), cond16[n * E:])
        L = spec.L
        ctx.interleaved = bool(runtime.CFG.kv_interleaved)
        if ctx.interleaved:
            # [rows][layer][K | V]: every layer's projection in one launch (N = L * 2 D); a layer is the view kv[li] with row stride L * 2 D
            kvbuf = _empty((sets * n * E, L * 2 * D), dt, dev)
            _fwd_gemm(cond16, "bkv_all", _kv_all_weights(kvparams, L), kvbuf, bias=torch.cat([kvparams[4 * li + j].detach() for li in range(L) for j in (1, 3)]))
            kv = None
        else:
            kv = _empty((L, sets * n * E, 2 * D), dt, dev)
            for li in range(L):
                wk, bk, wv, bv = kvparams[4 * li: 4 * li + 4]
                _fwd_gemm(cond16, "bkv", [wk, wv], kv[li], bias=torch.cat((bk.detach(), bv.detach())))
        ctx.spec, ctx.kvparams, ctx.cond16, ctx.shape, ctx.sets = spec, kvparams, cond16, (n, E, D), sets
        ctx.needs = (cond_own.requires_grad, cond_neg is not None and cond_neg.requires_grad)
        if ctx.interleaved:      # 2-D [rows, L * 2 D]: contiguous row blocks (so are their gradients: autograd's sums over the passes stay on its fast path)
            return (kvbuf[:n * E], kvbuf[n * E:]) if sets == 2 else (kvbuf, None)
        if sets == 2:
            return kv[:, :n * E], kv[:, n * E:]
        return kv, None

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dkv_own, dkv_neg):
        spec, kvparams, cond16 = ctx.spec, ctx.kvparams, ctx.cond16
        n, E, D = ctx.shape
        dev = cond16.device
        inv_s = 1.0 / runtime.grad_scale()
        if ctx.session is not None:
            sess = ctx.session
            # The session's contract: the buffer the first reader registered IS the gradient the engine hands over here - later readers
            # added to it in place and returned None.  Were it ever copied or cast on the way (a tensor hook, another producer stream's InputBuffer
            # path, a future engine), those additions would sit in an orphan buffer and the K/V weight and condition-token gradients would be short -
            # silently.  Checked by address; MICO_DKV_PER_PASS=1 is the way out.
            if sess.own is not None and not sess.closed and (dkv_own is None or dkv_own.data_ptr() != sess.own.data_ptr()
                                                             or tuple(dkv_own.shape) != tuple(sess.own.shape)):
                sess.reset()
                raise RuntimeError("CrossKVFn.backward: the gradient of the shared own K/V set is not the buffer its readers accumulated into "
                                   "(functional.DkvSession) - set MICO_DKV_PER_PASS=1 (one buffer per pass, summed by autograd)")
            sess.reset()
        parts = [(dkv_own, cond16[:n * E])]
        if ctx.sets == 2:
            parts.append((dkv_neg, cond16[n * E:]))
        L = spec.L
        if ctx.interleaved:
            # the gradients arrived in the forward's layout ([rows][layer][dK | dV], BertFn writes them so and autograd's sums keep it): all layers
            # at once - dW [L * 2 D, D] in one launch per set, the condition-token gradient as ONE product over K = L * 2 D
            CrossKVFn.concat_backwards += 1
            wall = _fused_w("bkv_all", _kv_all_weights(kvparams, L))
            dw = torch.zeros((L * 2 * D, D), dtype=torch.float32, device=dev)
            db = torch.zeros(L * 2 * D, dtype=torch.float32, device=dev)
            dconds = [None, None]
            for pi, (dkv, c16) in enumerate(parts):
                if dkv is None:
                    continue
                d2 = dkv if (dkv.stride(1) == 1 and dkv.stride(0) % 8 == 0) else dkv.contiguous()      # [n E, L * 2 D]
                # (the weight-gradient kernels address a reduction-major operand with 32-bit byte offsets from its first row: row chunks of < 4 GiB)
                rmax = max(64, (0xF0000000 // (L * 2 * D * d2.element_size())) // 64 * 64)
                for r0 in range(0, n * E, rmax):
                    linear_wgrad(d2[r0:r0 + rmax], c16[r0:r0 + rmax], dw, inv_s, dbias=db)
                if ctx.needs[pi]:
                    dconds[pi] = _empty((n * E, D), torch.float32, dev)
                    ops.gemm(d2, wall, dconds[pi], tb=True, M=n * E, N=D, K=L * 2 * D, alpha=inv_s)
            grads = []
            for li in range(L):
                r0 = li * 2 * D
                grads += [dw[r0:r0 + D], db[r0:r0 + D], dw[r0 + D:r0 + 2 * D], db[r0 + D:r0 + 2 * D]]
            dc = [d.view(n, E, D) if d is not None else None for d in dconds]
            return (None, None, dc[0], dc[1]) + tuple(grads)
        dconds = [torch.zeros((n * E, D), dtype=torch.float32, device=dev) if (need and d is not None) else None
                  for need, (d, _) in zip(ctx.needs, parts)] + [None] * (2 - len(parts))
        grads = []
        for li in range(spec.L):
            wk, bk, wv, bv = kvparams[4 * li: 4 * li + 4]
            dw = torch.zeros((2 * D, D), dtype=torch.float32, device=dev)
            db = torch.zeros(2 * D, dtype=torch.float32, device=dev)
            for pi, (dkv, c16) in enumerate(parts):
                if dkv is None:
                    continue
                d = dkv[li]
                linear_wgrad(d, c16, dw, inv_s)
                ops.colsum(d, db, scale=inv_s, accumulate=True)
                if dconds[pi] is not None:
                    ops.gemm(d, _fused_w("bkv", [wk, wv]), dconds[pi], tb=True, M=n * E, N=D, K=2 * D, alpha=inv_s, accumulate=True)
            grads += [dw[:D], db[:D], dw[D:], db[D:]]
        dc = [d.view(n, E, D) if d is not None else None for d in dconds]
        return (None, None, dc[0], dc[1]) + tuple(grads)


class BertFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, input_ids, add_mask, cond, drop, kv_own, kv_neg, *params):
        """kv_own / kv_neg (CrossKVFn outputs, [L][n E, 2 D]) instead of cond: the cross-attention K/V memory is given; a batch of
        n entries reads kv_own, one of 3 n entries is the ITM triplet [own | neg | own] and reads [kv_own | kv_neg] modulo 2 n.
        drop: None (eval), (p_hidden, p_attention, seed) - train-mode dropout of bert.py:148,267,295,373 - or a dict
        {"kv_cache": {...}} (inference only): the per-layer cross-attention K/V projections of `cond` are stored in / taken from
        that dict, so a decode loop projects its (constant) condition tokens once instead of at every step."""
        runtime.remember_precision(ctx)
        kv_cache = None
        if isinstance(drop, dict):
            kv_cache, drop = drop["kv_cache"], None
        dt = runtime.compute_dtype()
        ph, pa, dseed = drop if drop is not None else (0.0, 0.0, 0)
        hd_drop = (lambda site: (ph, dseed, site)) if ph > 0 else (lambda site: None)
        at_drop = (lambda site: (pa, dseed, site)) if pa > 0 else (lambda site: None)
        P = lambda n: params[spec.idx[n]]
        dev = input_ids.device
        b, S = input_ids.shape
        D, H, I = spec.D, spec.H, spec.I
        hd = D // H
        rows = b * S
        ids = input_ids.contiguous()
        emb = _empty((rows, D), torch.float32, dev)
        ops.bert_embed_fwd(ids, P("embeddings.word_embeddings.weight").detach(), P("embeddings.position_embeddings.weight").detach(),
                           P("embeddings.token_type_embeddings.weight").detach()[0].contiguous(), emb, S)
        split0 = runtime.split_activations()
        x32, x16 = _empty((rows, D), torch.float32, dev), _empty((rows, 2 * D if split0 else D), dt, dev)
        mean_e, rstd_e = _empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev)
        ops.layernorm_fwd(emb, P("embeddings.LayerNorm.weight"), P("embeddings.LayerNorm.bias"), spec.eps, out16=x16, out32=x32,
                          mean=mean_e, rstd=rstd_e, split16=split0, dtype=dt, drop=hd_drop(SITE_EMB))
        cond16 = None
        E = 0
        kv_mod = 0
        ctx.dkv_session = getattr(kv_own, "_mico_dkv", None) if kv_own is not None else None      # (DkvSession, attached by project_cross_kv)
        if kv_own is not None:
            assert cond is None and kv_cache is None
            n_own = b if kv_neg is None else b // 3
            kv_2d = kv_own.dim() == 2        # the interleaved memory: [n E, L * 2 D] row-major (CrossKVFn), read per layer through its row stride
            if kv_2d:
                assert kv_own.is_contiguous() and kv_own.shape[1] == spec.L * 2 * D
                kv_own = _kv_layers(kv_own, spec.L, D)
            E = kv_own.shape[1] // n_own
            if kv_neg is not None:   # the triplet reads one [own | neg] buffer modulo 2 n: the two views must be adjacent per layer
                if kv_2d:
                    assert kv_neg.dim() == 2 and kv_neg.is_contiguous()
                    kv_neg = _kv_layers(kv_neg, spec.L, D)
                assert b == 3 * n_own and kv_neg.stride() == kv_own.stride() and \
                    kv_neg.data_ptr() == kv_own.data_ptr() + n_own * E * kv_own.stride(1) * kv_own.element_size()
                kv_mod = 2 * n_own
        if cond is not None:
            E = cond.shape[1]
            cond16 = _empty((b * E, D), dt, dev)
            ops.cast_f32_to_16(cond.contiguous().view(b * E, D), cond16)
        mask = add_mask.contiguous() if add_mask is not None else None
        scale = 1.0 / math.sqrt(hd)
        acts = []

        split = runtime.split_activations()

        def ln_out(u, pre):
            o32 = _empty((rows, D), torch.float32, dev)
            o16 = _empty((rows, 2 * D if split else D), dt, dev)
            m_, r_ = _empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev)
            ops.layernorm_fwd(u, P(pre + "LayerNorm.weight"), P(pre + "LayerNorm.bias"), spec.eps, out16=o16, out32=o32, mean=m_, rstd=r_,
                              split16=split, dtype=dt)
            return o32, o16, m_, r_

        for li in range(spec.L):
            p = f"encoder.layer.{li}."
            a = dict(x16=x16[:, :D])
            sa = p + "attention.self."
            bqkv = torch.cat((P(sa + "query.bias").detach(), P(sa + "key.bias").detach(), P(sa + "value.bias").detach()))
            qkv = _empty((rows, 3 * D), dt, dev)
            _fwd_gemm(x16, "bqkv", [P(sa + "query.weight"), P(sa + "key.weight"), P(sa + "value.weight")], qkv, bias=bqkv)
            co = _empty((rows, D), dt, dev)
            lse = _empty((b, H, S), torch.float32, dev)
            st = dict(q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
            ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], co, lse, B=b, H=H, Sq=S, Sk=S, hd=hd, scale=scale, mask=mask,
                         drop=at_drop(li * 8 + SITE_SELF_P), **st)
            wo = P(p + "attention.output.dense.weight")
            u = _empty((rows, D), torch.float32, dev)
            _fwd_gemm(co, "w1", [wo], u, bias=P(p + "attention.output.dense.bias"), resid=x32, drop=hd_drop(li * 8 + SITE_SELF_OUT))
            x32, x16, m1, r1 = ln_out(u, p + "attention.output.")
            a.update(qkv=qkv, co=co, lse=lse, u=u, m1=m1, r1=r1)
            if cond16 is not None or kv_own is not None:
                ca = p + "crossattention.self."
                a["x16a"] = x16[:, :D]
                q = _empty((rows, D), dt, dev)
                _fwd_gemm(x16, "w1", [P(ca + "query.weight")], q, bias=P(ca + "query.bias"))
                kv = kv_cache.get(li) if kv_cache is not None else None
                if kv_own is not None:
                    kv = kv_own[li]      # with kv_mod: the first n E rows of the layer's [own | neg] buffer
                elif kv is None or kv.shape[0] != b * E:
                    bkv = torch.cat((P(ca + "key.bias").detach(), P(ca + "value.bias").detach()))
                    kv = _empty((b * E, 2 * D), dt, dev)
                    _fwd_gemm(cond16, "bkv", [P(ca + "key.weight"), P(ca + "value.weight")], kv, bias=bkv)
                    if kv_cache is not None:
                        kv_cache[li] = kv
                cc = _empty((rows, D), dt, dev)
                lse_c = _empty((b, H, S), torch.float32, dev)
                krs = kv.stride(0)       # 2 D, or L * 2 D for a layer of the interleaved shared memory (CrossKVFn)
                stc = dict(q_strides=(S * D, D), k_strides=(E * krs, krs), v_strides=(E * krs, krs), o_strides=(S * D, D))
                ops.attn_fwd(q, kv, kv[:, D:], cc, lse_c, B=b, H=H, Sq=S, Sk=E, hd=hd, scale=scale, mask=None,
                             drop=at_drop(li * 8 + SITE_CROSS_P), kv_batch_mod=kv_mod, **stc)
                u2 = _empty((rows, D), torch.float32, dev)
                _fwd_gemm(cc, "w1", [P(p + "crossattention.output.dense.weight")], u2,
                          bias=P(p + "crossattention.output.dense.bias"), resid=x32, drop=hd_drop(li * 8 + SITE_CROSS_OUT))
                x32, x16, m2, r2 = ln_out(u2, p + "crossattention.output.")
                a.update(q=q, kv=kv, cc=cc, lse_c=lse_c, u2=u2, m2=m2, r2=r2)
            a["x16b"] = x16[:, :D]
            h = _empty((rows, I), dt, dev)
            act = _empty((rows, I), dt, dev)
            _fwd_gemm(x16, "w1", [P(p + "intermediate.dense.weight")], act, bias=P(p + "intermediate.dense.bias"),
                      aux_out=h, act=ops.ACT_GELU_SAVE_DERIV)
            u3 = _empty((rows, D), torch.float32, dev)
            _fwd_gemm(act, "w1", [P(p + "output.dense.weight")], u3, bias=P(p + "output.dense.bias"), resid=x32,
                      drop=hd_drop(li * 8 + SITE_FFN_OUT))
            x32, x16, m3, r3 = ln_out(u3, p + "output.")
            a.update(h=h, act=act, u3=u3, m3=m3, r3=r3)
            acts.append(a)
        ctx.spec, ctx.params, ctx.acts, ctx.dt = spec, params, acts, dt
        ctx.misc = (ids, emb, mean_e, rstd_e, cond16, mask, b, S, E)
        ctx.drop = drop
        ctx.cond_needs_grad = cond is not None and cond.requires_grad
        ctx.kv_shared = (kv_own is not None, kv_neg is not None, kv_mod)
        ctx.kv_2d = kv_own is not None and kv_2d
        return x32.view(b, S, D)

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dseq):
        spec, params, dt = ctx.spec, ctx.params, ctx.dt
        P = lambda n: params[spec.idx[n]]
        ids, emb, mean_e, rstd_e, cond16, mask, b, S, E = ctx.misc
        ph, pa, dseed = ctx.drop if ctx.drop is not None else (0.0, 0.0, 0)
        at_drop = (lambda site: (pa, dseed, site)) if pa > 0 else (lambda site: None)
        dev = dseq.device
        D, H, I = spec.D, spec.H, spec.I
        hd = D // H
        rows = b * S
        Sg = runtime.grad_scale()
        inv_s = 1.0 / Sg
        scale = 1.0 / math.sqrt(hd)
        # private = parameters with gradient producers OUTSIDE the BertFn nodes of this backward pass: the word embeddings (tied to the LM
        # head's decoder) and - in a pass that projects its own condition tokens while the step may also hold a CrossKVFn node for the same
        # key / value projections (share_cross_kv: e.g. "cap%tv%ta_ret%tva", where cap%ta has no retrieval twin) - the cross-attention
        # key / value weights and biases.  Two defined gradients for one parameter make the engine replace its accumulator by an
        # out-of-place sum; in-place additions to an arena view handed out earlier would then be lost.
        private = [spec.idx["embeddings.word_embeddings.weight"]]
        if cond16 is not None and runtime.CFG.share_cross_kv:
            private += [i for grp in spec.grad_groups for i in grp if ".crossattention.self." in spec.names[i]]
        grads = GradArena.session(params, spec.grad_groups, private=private)

        def G(name):
            return grads.get(spec.idx[name])

        def GF(names, shape):   # the adjacent gradient views of a fused projection as one matrix / vector
            return grads.fused([spec.idx[n] for n in names], shape)

        g = dseq.contiguous().view(rows, D).float().clone()
        dcond = torch.zeros((b * E, D), dtype=torch.float32, device=dev) if cond16 is not None else None
        shared, has_neg, kv_mod = ctx.kv_shared
        n_own = (b // 3 if has_neg else b) if shared else 0
        # gradients of the shared K/V memory, still in the 16-bit gradient scale (CrossKVFn.backward removes it)
        # (ITM triplet: [own | neg] adjacent per layer - batch entries 0 .. 2 n of the backward write their dK / dV rows straight into the pair, the
        # third third accumulates onto the own half in a second launch: no per-entry buffer, no add / copy passes)
        kv_il = shared and ctx.kv_2d      # the memory is interleaved over the layers (a 2-D tensor to autograd): so are its gradients
        dkv2d = None        # interleaved: the 2-D [rows, L * 2 D] buffer behind the per-layer views - what autograd gets back
        # DkvSession: the passes that read one own set share ONE gradient buffer for it.  acc_own = the buffer an earlier backward of this pass
        # wrote (this one adds to it in the attention kernel and returns None for kv_own).  Only when the engine wants kv_own's gradient in this
        # pass at all - then CrossKVFn.backward runs after every reader and ends the session.
        # (and only under an engine that can tell backward passes apart - the condition GradArena.session has: without it, a buffer per pass)
        sess = ctx.dkv_session if (kv_il and runtime.CFG.dkv_inplace and ctx.needs_input_grad[5] and hasattr(torch._C, "_current_graph_task_id")) else None
        acc_own = None
        if sess is not None and sess.own is not None and not sess.closed:
            if (ops.attn_bwd_smallq_ok(n_own, H, S, E, hd, at_drop(SITE_CROSS_P), batch0=2 * n_own if has_neg else 0)
                    and tuple(sess.own.shape) == (n_own * E, spec.L * 2 * D) and sess.own.dtype == dt):
                acc_own = sess.own
            else:
                sess.closed = True
        dkv2d_neg = None
        if acc_own is not None:
            dkv_pair = None
            dkv_own = _kv_layers(acc_own, spec.L, D)
            dkv_neg = None
            if has_neg:        # the triplet's hard negatives: only this pass reads them - a buffer of their own
                dkv2d_neg = _empty((n_own * E, spec.L * 2 * D), dt, dev)
                dkv_neg = _kv_layers(dkv2d_neg, spec.L, D)
        elif has_neg:
            if kv_il:
                dkv2d = _empty((2 * n_own * E, spec.L * 2 * D), dt, dev)
                dkv_pair = _kv_layers(dkv2d, spec.L, D)
            else:
                dkv_pair = _empty((spec.L, 2 * n_own * E, 2 * D), dt, dev)
            dkv_own, dkv_neg = dkv_pair[:, :n_own * E], dkv_pair[:, n_own * E:]
        else:
            dkv_pair = None
            dkv_neg = None
            if shared and kv_il:
                dkv2d = _empty((n_own * E, spec.L * 2 * D), dt, dev)
                dkv_own = _kv_layers(dkv2d, spec.L, D)
            else:
                dkv_own = _empty((spec.L, n_own * E, 2 * D), dt, dev) if shared else None

        def ln_bwd(gin, u, m_, r_, pre, site):
            """d(LN input) fp32 (in place into gin: the residual branch's gradient) and its scaled 16-bit copy for the dense
            branch, which sat behind a dropout in training: the same mask multiplies its gradient."""
            d16 = _empty((rows, D), dt, dev)
            ops.layernorm_bwd(gin, u, P(pre + "LayerNorm.weight"), m_, r_, dx32=gin, dx16=d16, scale16=Sg,
                              dgamma=G(pre + "LayerNorm.weight"), dbeta=G(pre + "LayerNorm.bias"), dtype=dt,
                              dx16_drop=(ph, dseed, site) if ph > 0 else None)
            return d16

        for li in reversed(range(spec.L)):
            p = f"encoder.layer.{li}."
            a = ctx.acts.pop()
            # ---- FFN ----
            d16 = ln_bwd(g, a["u3"], a["m3"], a["r3"], p + "output.", li * 8 + SITE_FFN_OUT)   # g := dL/du3 (= dL/d(resid) too)
            linear_wgrad(d16, a["act"], G(p + "output.dense.weight"), inv_s)
            ops.colsum(d16, G(p + "output.dense.bias"), scale=inv_s, accumulate=True)
            dh = a["act"]
            ops.gemm(d16, _fused_w("w1", [P(p + "output.dense.weight")]), dh, tb=True, M=rows, N=I, K=D, aux_in=a["h"], act=ops.ACT_MUL_AUX)
            linear_wgrad(dh, a["x16b"], G(p + "intermediate.dense.weight"), inv_s)
            ops.colsum(dh, G(p + "intermediate.dense.bias"), scale=inv_s, accumulate=True)
            ops.gemm(dh, _fused_w("w1", [P(p + "intermediate.dense.weight")]), g, tb=True, M=rows, N=D, K=I, alpha=inv_s, resid=g)
            # ---- cross attention ----
            if cond16 is not None or shared:
                ca = p + "crossattention.self."
                d16 = ln_bwd(g, a["u2"], a["m2"], a["r2"], p + "crossattention.output.", li * 8 + SITE_CROSS_OUT)
                linear_wgrad(d16, a["cc"], G(p + "crossattention.output.dense.weight"), inv_s)
                ops.colsum(d16, G(p + "crossattention.output.dense.bias"), scale=inv_s, accumulate=True)
                dcc = _empty((rows, D), dt, dev)
                ops.gemm(d16, _fused_w("w1", [P(p + "crossattention.output.dense.weight")]), dcc, tb=True, M=rows, N=D, K=D)
                dq = _empty((rows, D), dt, dev)
                delta = _empty((b, H, S), torch.float32, dev)
                kv = a["kv"]
                two_launch = shared and has_neg and ops.attn_bwd_smallq_ok(n_own, H, S, E, hd, at_drop(li * 8 + SITE_CROSS_P), batch0=2 * n_own)
                if shared and has_neg and not two_launch and kv.stride(0) != 2 * D:
                    # one-launch triplet backward into a per-entry [3 n E, 2 D] buffer: dK / dV use the K / V strides, so this (rare) path reads a
                    # row-major copy of the layer's [own | neg] sets
                    kv = torch.as_strided(kv, (2 * n_own * E, 2 * D), (kv.stride(0), 1)).contiguous()
                krs = kv.stride(0)
                stc = dict(q_strides=(S * D, D), k_strides=(E * krs, krs), v_strides=(E * krs, krs), o_strides=(S * D, D))
                if two_launch and acc_own is None:
                    assert dkv_pair[li].stride(0) == krs
                if acc_own is not None and has_neg:
                    # three launches over the triplet: entries [0, n) and [2 n, 3 n) ADD their dK / dV to the session's own-set buffer (the captioning
                    # pass wrote it), entries [n, 2 n) write the hard negatives' buffer
                    ne, r1 = n_own * E, n_own * S
                    dko, dkn = dkv_own[li], dkv_neg[li]
                    assert dko.stride(0) == krs and dkn.stride(0) == krs
                    kvn = torch.as_strided(kv, kv.shape, kv.stride(), kv.storage_offset() + ne * krs)     # (kv = the own rows of the layer's [own | neg] sets)
                    for e0, kvs, dst, acc in ((0, kv, dko, True), (1, kvn, dkn, False), (2, kv, dko, True)):
                        rs = slice(e0 * r1, (e0 + 1) * r1)
                        ops.attn_bwd(a["q"][rs], kvs, kvs[:, D:], a["cc"][rs], dcc[rs], a["lse_c"][e0 * n_own:(e0 + 1) * n_own], dq[rs], dst, dst[:, D:],
                                     delta, B=n_own, H=H, Sq=S, Sk=E, hd=hd, scale=scale, mask=None, drop=at_drop(li * 8 + SITE_CROSS_P),
                                     batch0=e0 * n_own, dkv_accumulate=acc, **stc)
                    dkv = None
                    split_done = True
                elif shared and has_neg and ops.attn_bwd_smallq_ok(n_own, H, S, E, hd, at_drop(li * 8 + SITE_CROSS_P), batch0=2 * n_own):
                    # two launches over the triplet: entries [0, 2 n) own exactly the [own | neg] K/V sets, entries [2 n, 3 n) read the own sets again
                    # and ADD their dK / dV (mico_attn_params.batch0 keeps the dropout counters of the one-launch forward)
                    dkv = dkv_pair[li]
                    r2 = 2 * n_own * S
                    ops.attn_bwd(a["q"][:r2], kv, kv[:, D:], a["cc"][:r2], dcc[:r2], a["lse_c"][:2 * n_own], dq[:r2], dkv, dkv[:, D:], delta,
                                 B=2 * n_own, H=H, Sq=S, Sk=E, hd=hd, scale=scale, mask=None, drop=at_drop(li * 8 + SITE_CROSS_P), **stc)
                    ops.attn_bwd(a["q"][r2:], kv, kv[:, D:], a["cc"][r2:], dcc[r2:], a["lse_c"][2 * n_own:], dq[r2:], dkv, dkv[:, D:], delta,
                                 B=n_own, H=H, Sq=S, Sk=E, hd=hd, scale=scale, mask=None, drop=at_drop(li * 8 + SITE_CROSS_P), batch0=2 * n_own,
                                 dkv_accumulate=True, **stc)
                    split_done = True
                else:
                    split_done = False
                    dkv = dkv_own[li] if (shared and not has_neg) else _empty((b * E, 2 * D), dt, dev)
                    ops.attn_bwd(a["q"], kv, kv[:, D:], a["cc"], dcc, a["lse_c"], dq, dkv, dkv[:, D:], delta, B=b, H=H, Sq=S, Sk=E,
                                 hd=hd, scale=scale, mask=None, drop=at_drop(li * 8 + SITE_CROSS_P), kv_batch_mod=kv_mod,
                                 dkv_accumulate=acc_own is not None, **stc)
                linear_wgrad(dq, a["x16a"], G(ca + "query.weight"), inv_s)
                ops.colsum(dq, G(ca + "query.bias"), scale=inv_s, accumulate=True)
                if shared:   # dK/dV per batch entry -> per K/V set: the triplet's first and third thirds read the same (own) set
                    ne = n_own * E
                    if has_neg and not split_done:
                        torch.add(dkv[:ne], dkv[2 * ne:], out=dkv_own[li])
                        dkv_neg[li].copy_(dkv[ne:2 * ne])
                else:
                    linear_wgrad(dkv, cond16, GF([ca + "key.weight", ca + "value.weight"], (2 * D, D)), inv_s)
                    ops.colsum(dkv, GF([ca + "key.bias", ca + "value.bias"], (2 * D,)), scale=inv_s, accumulate=True)
                if ctx.cond_needs_grad:
                    wkv = _fused_w("bkv", [P(ca + "key.weight"), P(ca + "value.weight")])
                    ops.gemm(dkv, wkv, dcond, tb=True, M=b * E, N=D, K=2 * D, alpha=inv_s, accumulate=True)
                ops.gemm(dq, _fused_w("w1", [P(ca + "query.weight")]), g, tb=True, M=rows, N=D, K=D, alpha=inv_s, resid=g)
                del dq, dkv, dcc
            # ---- self attention ----
            sa = p + "attention.self."
            d16 = ln_bwd(g, a["u"], a["m1"], a["r1"], p + "attention.output.", li * 8 + SITE_SELF_OUT)
            linear_wgrad(d16, a["co"], G(p + "attention.output.dense.weight"), inv_s)
            ops.colsum(d16, G(p + "attention.output.dense.bias"), scale=inv_s, accumulate=True)
            dco = _empty((rows, D), dt, dev)
            ops.gemm(d16, _fused_w("w1", [P(p + "attention.output.dense.weight")]), dco, tb=True, M=rows, N=D, K=D)
            qkv = a["qkv"]
            dqkv = _empty((rows, 3 * D), dt, dev)
            delta = _empty((b, H, S), torch.float32, dev)
            st = dict(q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
            ops.attn_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], a["co"], dco, a["lse"], dqkv, dqkv[:, D:], dqkv[:, 2 * D:], delta,
                         B=b, H=H, Sq=S, Sk=S, hd=hd, scale=scale, mask=mask, drop=at_drop(li * 8 + SITE_SELF_P), **st)
            linear_wgrad(dqkv, a["x16"], GF([sa + "query.weight", sa + "key.weight", sa + "value.weight"], (3 * D, D)), inv_s)
            ops.colsum(dqkv, GF([sa + "query.bias", sa + "key.bias", sa + "value.bias"], (3 * D,)), scale=inv_s, accumulate=True)
            wqkv = _fused_w("bqkv", [P(sa + "query.weight"), P(sa + "key.weight"), P(sa + "value.weight")])
            ops.gemm(dqkv, wqkv, g, tb=True, M=rows, N=D, K=3 * D, alpha=inv_s, resid=g)
            del a, dqkv, dco, d16
        # ---- embeddings ----
        if ph > 0:
            ops.dropout_(g, (ph, dseed, SITE_EMB))
        ops.layernorm_bwd(g, emb, P("embeddings.LayerNorm.weight"), mean_e, rstd_e, dx32=g,
                          dgamma=G("embeddings.LayerNorm.weight"), dbeta=G("embeddings.LayerNorm.bias"), dtype=dt)
        dtype0 = torch.zeros(D, dtype=torch.float32, device=dev)
        ops.embed_scatter_add(ids, g, G("embeddings.word_embeddings.weight"), G("embeddings.position_embeddings.weight"), dtype0, S)
        G("embeddings.token_type_embeddings.weight")[0].add_(dtype0)
        dc = dcond.view(b, E, D) if (dcond is not None and ctx.cond_needs_grad) else None
        if dkv2d is not None:      # contiguous row blocks [own | neg] of the one buffer
            dkv_own, dkv_neg = (dkv2d[:n_own * E], dkv2d[n_own * E:]) if has_neg else (dkv2d, None)
        if acc_own is not None:    # the own set's gradient is in the buffer another pass handed to autograd
            dkv_own, dkv_neg = None, dkv2d_neg
            DkvSession.accumulated += 1
        elif sess is not None:
            if sess.own is None and not sess.closed and dkv2d is not None:
                sess.own = dkv_own
            else:
                sess.closed = True      # a second buffer for the same set: the engine sums out of place from here on
        return (None, None, None, dc, None, dkv_own, dkv_neg) + grads.result()


# ======================================================================================================================
# LM head + masked-token cross entropy (bert.py:575-609, 1085-1090): dense+GELU+LN -> tied decoder GEMM -> CE, fused so
# that fp32 logits never exist; only labelled rows are pushed through the 768 x 30522 GEMM (identical result: ignored
# rows contribute neither loss nor gradient).
# ======================================================================================================================
VOCAB_PAD = 64


class LMHeadLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seq, labels, wt, bt, g, beta, wdec, bdec):
        runtime.remember_precision(ctx)
        dt = runtime.compute_dtype()
        dev = seq.device
        D = seq.shape[-1]
        x = seq.reshape(-1, D).contiguous().float()
        lab = labels.reshape(-1).contiguous()
        rows = x.shape[0]
        V = wdec.shape[0]
        Vp = (V + VOCAB_PAD - 1) // VOCAB_PAD * VOCAB_PAD
        x16 = _empty((rows, D), dt, dev)
        ops.cast_f32_to_16(x, x16)
        pre = _empty((rows, D), dt, dev)
        act = _empty((rows, D), dt, dev)
        _fwd_gemm(x16, "w1", [wt], act, bias=bt.detach(), aux_out=pre, act=ops.ACT_GELU)
        hl = _empty((rows, D), dt, dev)
        mean, rstd = _empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev)
        ops.layernorm_fwd(act, g.detach(), beta.detach(), 1e-12, out16=hl, mean=mean, rstd=rstd, dtype=dt)
        wd16, ksd = runtime.gemm_weight([wdec], "wdec", n_pad=Vp)
        bpad = torch.zeros(Vp, dtype=torch.float32, device=dev)
        bpad[:V] = bdec.detach()
        logits = _empty((rows, Vp), dt, dev)
        ops.gemm(hl, wd16, logits, bias=bpad, ksegs=ksd)
        row_loss = _empty((rows,), torch.float32, dev)
        ops.ce_fwd_bwd(logits, lab, cols=V, row_loss=row_loss)
        n_valid = (lab != -100).sum().clamp_min(1).float()
        loss = row_loss.sum() / n_valid
        ctx.save_for_backward(x16, pre, act, hl, mean, rstd, logits, lab, n_valid, wt, g, wdec)
        ctx.meta = (dt, V, Vp, seq.shape)
        return loss

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, gout):
        x16, pre, act, hl, mean, rstd, logits, lab, n_valid, wt, g, wdec = ctx.saved_tensors
        dt, V, Vp, seq_shape = ctx.meta
        dev = gout.device
        rows, D = x16.shape
        S = runtime.grad_scale()
        inv_s = 1.0 / S
        dsc = (gout.float() * S / n_valid).reshape(1).contiguous()
        if getattr(ctx, "_logits_consumed", False):
            raise RuntimeError("LMHeadLossFn: a second backward through the same graph - the 16-bit logits were overwritten by their "
                               "gradient in the first one (no fp32 [rows, 30522] copy is kept); re-run the forward instead of retain_graph")
        ctx._logits_consumed = True
        dlog = logits   # overwrite the logits with their gradient (same dtype / shape)
        if Vp != V:
            dlog[:, V:].zero_()
        ops.ce_fwd_bwd(logits, lab, cols=V, dlogits=dlog, dscale_ptr=dsc)
        dwdec = torch.zeros(wdec.shape, dtype=torch.float32, device=dev)
        linear_wgrad(dlog, hl, dwdec, inv_s, n_out=V, n_in=D)
        dbdec = torch.zeros(V, dtype=torch.float32, device=dev)
        ops.colsum(dlog, dbdec, cols=V, scale=inv_s)
        wd16 = runtime.gemm_weight([wdec], "wdec", n_pad=Vp)[0]
        dhl = _empty((rows, D), dt, dev)
        ops.gemm(dlog, wd16, dhl, tb=True, M=rows, N=D, K=Vp)
        dact = _empty((rows, D), dt, dev)
        dg, dbeta = torch.zeros_like(g, dtype=torch.float32), torch.zeros_like(g, dtype=torch.float32)
        ops.layernorm_bwd(dhl, act, g.detach(), mean, rstd, dx16=dact, dgamma=dg, dbeta=dbeta, grad_scale=inv_s, dtype=dt)
        dpre = dhl
        ops.gelu_bwd_16(pre, dact, dpre)
        dwt = torch.zeros(wt.shape, dtype=torch.float32, device=dev)
        linear_wgrad(dpre, x16, dwt, inv_s)
        dbt = torch.zeros(D, dtype=torch.float32, device=dev)
        ops.colsum(dpre, dbt, scale=inv_s)
        dx = _empty((rows, D), torch.float32, dev)
        ops.gemm(dpre, _fused_w("w1", [wt]), dx, tb=True, M=rows, N=D, K=D, alpha=inv_s)
        return dx.view(seq_shape), None, dwt, dbt, dg, dbeta, dwdec, dbdec


class LMLogitsFn(torch.autograd.Function):
    """Inference-only LM logits (bert.py:1085): fp32 [rows, V] out; no backward (training uses LMHeadLossFn)."""

    @staticmethod
    def forward(ctx, seq, wt, bt, g, beta, wdec, bdec):
        dt = runtime.compute_dtype()
        dev = seq.device
        D = seq.shape[-1]
        x = seq.reshape(-1, D).contiguous().float()
        rows = x.shape[0]
        V = wdec.shape[0]
        Vp = (V + VOCAB_PAD - 1) // VOCAB_PAD * VOCAB_PAD
        x16 = _empty((rows, D), dt, dev)
        ops.cast_f32_to_16(x, x16)
        act = _empty((rows, D), dt, dev)
        _fwd_gemm(x16, "w1", [wt], act, bias=bt.detach(), act=ops.ACT_GELU)
        hl = _empty((rows, D), dt, dev)
        ops.layernorm_fwd(act, g.detach(), beta.detach(), 1e-12, out16=hl, dtype=dt)
        wd16, ksd = runtime.gemm_weight([wdec], "wdec", n_pad=Vp)
        bpad = torch.zeros(Vp, dtype=torch.float32, device=dev)
        bpad[:V] = bdec.detach()
        logits = _empty((rows, Vp), torch.float32, dev)
        ops.gemm(hl, wd16, logits, bias=bpad, ksegs=ksd)
        ctx.mark_non_differentiable(logits)
        return logits[:, :V].reshape(*seq.shape[:-1], V)

    @staticmethod
    def backward(ctx, *a):
        raise RuntimeError("LMLogitsFn is inference-only; use labels=... to train the captioning head")


# ======================================================================================================================
# cross entropy over small fp32 logits (ITC with label smoothing, ITM)  - vast.py:411-415, 455
# ======================================================================================================================
class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, label_smoothing):
        logits = logits.contiguous().float()
        rows = logits.shape[0]
        row_loss = _empty((rows,), torch.float32, logits.device)
        ops.ce_fwd_bwd(logits, target.contiguous(), label_smoothing=label_smoothing, row_loss=row_loss)
        n_valid = (target != -100).sum().clamp_min(1).float()
        ctx.save_for_backward(logits, target, n_valid)
        ctx.ls = label_smoothing
        return row_loss.sum() / n_valid

    @staticmethod
    def backward(ctx, gout):
        logits, target, n_valid = ctx.saved_tensors
        d = torch.empty_like(logits)
        dsc = (gout.float() / n_valid).reshape(1).contiguous()
        ops.ce_fwd_bwd(logits, target.contiguous(), label_smoothing=ctx.ls, dlogits=d, dscale_ptr=dsc)
        return d, None, None


def cross_entropy(logits, target, label_smoothing=0.0):
    return CrossEntropyFn.apply(logits, target, label_smoothing)
