"""Hand-written forward / backward schedule of the Swin tower on libmico_hip.so.

Reference: model/swin.py - PatchEmbed :437-475, SwinTransformerBlock :175-294 (WindowAttention :77-156, Mlp :26-42), PatchMerging
:315-352, BasicLayer :364-423, SwinTransformer.forward_features :588-600.  Same data flow as the EVA tower (functional.EvaTowerFn): the
residual stream and its gradient are fp32 [batch * tokens, C] and only ever touched by GEMM / LayerNorm epilogues, GEMM operands are 16-bit,
parameter gradients land in one flat fp32 arena.  What is specific to Swin:
  * window partition, cyclic shift, their inverses and the shift mask exist only as index arithmetic inside mico_win_attn_fwd / _bwd
    (csrc/swin.hip) - the qkv GEMM's output is consumed in token order and the attention output is written in token order;
  * the relative-position bias is read from (and its gradient accumulated into) the [169, heads] parameter table directly;
  * PatchMerging = mico_patch_merge (2x2 gather on the fp32 stream) + LayerNorm(4C) + bias-free GEMM.
Stochastic depth (DropPath, swin.py:218,291,294) multiplies a sample's branch by 0 or 1/keep in the output-projection epilogue
(row_scale); this tower evaluates dropped branches (no frame skipping - it is not on the timed path).
"""
import torch

from . import ops, runtime
from .functional import GradArena, _empty, _gemm_dx, _gemm_fwd, _ln16, linear_wgrad

EPS = 1e-5   # nn.LayerNorm default, swin.py:513 norm_layer=nn.LayerNorm


class SwinSpec:
    """Static description of one tower: names = parameter names in registration order (the order of `params` everywhere below)."""

    def __init__(self, names, img_size, patch, embed_dim, depths, heads, window, patch_norm=True, in_chans=3):
        assert window == 7, "the window-attention kernels are built for 7x7 windows (every Swin-T/S/B/L 224 configuration)"
        assert (img_size // patch) % 7 == 0 and all((embed_dim << i) // h == 32 for i, h in enumerate(heads)), \
            "grid side must be a multiple of 7 and the head dim 32"
        self.names = names
        self.idx = {n: i for i, n in enumerate(names)}
        self.img, self.P, self.C0, self.depths, self.heads, self.patch_norm, self.in_chans = img_size, patch, embed_dim, depths, heads, patch_norm, in_chans
        self.res0 = img_size // patch
        self.kpad = (in_chans * patch * patch + 63) // 64 * 64
        self.n_blocks = sum(depths)
        self.out_dim = embed_dim << (len(depths) - 1)
        self.out_tokens = (self.res0 >> (len(depths) - 1)) ** 2

    def stages(self):
        """-> (stage index, C, res, depth, heads, has_downsample)"""
        for s, d in enumerate(self.depths):
            yield s, self.C0 << s, self.res0 >> s, d, self.heads[s], s < len(self.depths) - 1

    @staticmethod
    def shift_of(res, j):
        return 3 if (j % 2 == 1 and res > 7) else 0   # swin.py:403 shift_size = window // 2 on odd blocks; :206-209 none when one window


def _swin_forward(spec, pixels, dp_scale, params, save):
    dt = runtime.compute_dtype()
    P = lambda n: params[spec.idx[n]]
    dev = params[0].device
    B = pixels.shape[0]
    # ---- patch embedding: im2row + GEMM(+bias), then LayerNorm on the fp32 stream (patch_norm) ----
    L0 = spec.res0 * spec.res0
    rows16 = _empty((B * L0, spec.kpad), dt, dev)
    ops.im2row(pixels.contiguous().float(), rows16, spec.P, spec.kpad)
    w16, ks = runtime.gemm_weight([P("patch_embed.proj.weight")], "swin_pe", k_pad=spec.kpad)
    x = _empty((B * L0, spec.C0), torch.float32, dev)
    ops.gemm(rows16, w16, x, M=B * L0, N=spec.C0, K=spec.kpad, bias=P("patch_embed.proj.bias"), ksegs=ks)
    pe = dict(rows16=rows16 if save else None)
    if spec.patch_norm:
        xn = _empty((B * L0, spec.C0), torch.float32, dev)
        mean, rstd = _empty((B * L0,), torch.float32, dev), _empty((B * L0,), torch.float32, dev)
        ops.layernorm_fwd(x, P("patch_embed.norm.weight"), P("patch_embed.norm.bias"), EPS, out32=xn, mean=mean, rstd=rstd, dtype=dt)
        pe.update(x=x if save else None, mean=mean, rstd=rstd)
        x = xn
    acts, merges = [], []
    bi = 0
    for s, C, res, depth, heads, down in spec.stages():
        L = res * res
        M = B * L
        for j in range(depth):
            b = f"layers.{s}.blocks.{j}."
            shift = SwinSpec.shift_of(res, j)
            sc1 = dp_scale[bi, 0].contiguous() if dp_scale is not None else None
            sc2 = dp_scale[bi, 1].contiguous() if dp_scale is not None else None
            ln1b, ln1, mean1, rstd1 = _ln16(x, P(b + "norm1.weight"), P(b + "norm1.bias"), EPS, M, C, dt, dev)
            qkv = _empty((M, 3 * C), dt, dev)
            _gemm_fwd(ln1b, C, [P(b + "attn.qkv.weight")], "w", qkv, bias=P(b + "attn.qkv.bias"))
            ao = _empty((M, C), dt, dev)
            lse = _empty((M, heads), torch.float32, dev)
            ops.win_attn_fwd(qkv, ao, lse, P(b + "attn.relative_position_bias_table").detach(), B, res, heads, shift, 32 ** -0.5)
            x_mid = _empty((M, C), torch.float32, dev)
            _gemm_fwd(ao, C, [P(b + "attn.proj.weight")], "w", x_mid, bias=P(b + "attn.proj.bias"), resid=x, row_scale=sc1, rows_per_scale=L)
            ln2b, ln2, mean2, rstd2 = _ln16(x_mid, P(b + "norm2.weight"), P(b + "norm2.bias"), EPS, M, C, dt, dev)
            h = _empty((M, 4 * C), dt, dev)
            act = _empty((M, 4 * C), dt, dev)
            _gemm_fwd(ln2b, C, [P(b + "mlp.fc1.weight")], "w", act, bias=P(b + "mlp.fc1.bias"), aux_out=h, act=ops.ACT_GELU_SAVE_DERIV)
            x_out = _empty((M, C), torch.float32, dev)
            _gemm_fwd(act, 4 * C, [P(b + "mlp.fc2.weight")], "w", x_out, bias=P(b + "mlp.fc2.bias"), resid=x_mid, row_scale=sc2, rows_per_scale=L)
            if save:
                acts.append(dict(x1=x, mean1=mean1, rstd1=rstd1, ln1=ln1, qkv=qkv, ao=ao, lse=lse, x2=x_mid, mean2=mean2, rstd2=rstd2, ln2=ln2,
                                 h=h, act=act, sc1=sc1, sc2=sc2))
            x = x_out
            bi += 1
        if down:
            d = f"layers.{s}.downsample."
            merged = _empty((M // 4, 4 * C), torch.float32, dev)
            ops.patch_merge(x, merged, B, res, C)
            lnb, ln, mean, rstd = _ln16(merged, P(d + "norm.weight"), P(d + "norm.bias"), EPS, M // 4, 4 * C, dt, dev)
            xn = _empty((M // 4, 2 * C), torch.float32, dev)
            _gemm_fwd(lnb, 4 * C, [P(d + "reduction.weight")], "w", xn)
            if save:
                merges.append(dict(merged=merged, ln=ln, mean=mean, rstd=rstd))
            x = xn
    Mo = B * spec.out_tokens
    out = _empty((Mo, spec.out_dim), torch.float32, dev)
    mean_n, rstd_n = _empty((Mo,), torch.float32, dev), _empty((Mo,), torch.float32, dev)
    ops.layernorm_fwd(x, P("norm.weight"), P("norm.bias"), EPS, out32=out, mean=mean_n, rstd=rstd_n, dtype=dt)
    saved = dict(acts=acts, merges=merges, pe=pe, final=(x, mean_n, rstd_n), dt=dt, B=B) if save else None
    return out.view(B, spec.out_tokens, spec.out_dim), saved


def _swin_backward(spec, params, saved, dout, grads):
    dt = saved["dt"]
    P = lambda n: params[spec.idx[n]]
    G = lambda n: grads.get(spec.idx[n])
    dev = dout.device
    B = saved["B"]
    S = runtime.grad_scale()
    inv_s = 1.0 / S
    x_last, mean_n, rstd_n = saved["final"]
    Mo = B * spec.out_tokens
    g = _empty((Mo, spec.out_dim), torch.float32, dev)      # gradient of the fp32 residual stream (unscaled)
    ops.layernorm_bwd(dout.contiguous().view(Mo, spec.out_dim), x_last, P("norm.weight"), mean_n, rstd_n, dx32=g,
                      dgamma=G("norm.weight"), dbeta=G("norm.bias"), dtype=dt)
    saved["final"] = None
    for s, C, res, depth, heads, down in reversed(list(spec.stages())):
        L = res * res
        M = B * L
        if down:
            d = f"layers.{s}.downsample."
            m = saved["merges"].pop()
            g16 = _empty((M // 4, 2 * C), dt, dev)
            ops.gather_rows_cast(g, g16, scale=S)
            linear_wgrad(g16, m["ln"], G(d + "reduction.weight"), inv_s)
            dln = _empty((M // 4, 4 * C), torch.float32, dev)
            _gemm_dx(g16, [P(d + "reduction.weight")], "w", dln)
            dmerged = _empty((M // 4, 4 * C), torch.float32, dev)
            ops.layernorm_bwd(dln, m["merged"], P(d + "norm.weight"), m["mean"], m["rstd"], dy_scale=inv_s, dx32=dmerged,
                              dgamma=G(d + "norm.weight"), dbeta=G(d + "norm.bias"), dtype=dt)
            g = _empty((M, C), torch.float32, dev)
            ops.patch_merge(dmerged, g, B, res, C, backward=True)
            del m, g16, dln, dmerged
        for j in reversed(range(depth)):
            b = f"layers.{s}.blocks.{j}."
            a = saved["acts"].pop()
            shift = SwinSpec.shift_of(res, j)
            # ---- MLP branch ----
            g16 = _empty((M, C), dt, dev)
            ops.gather_rows_cast(g, g16, row_scale=a["sc2"], rows_per_scale=L, scale=S)
            linear_wgrad(g16, a["act"], G(b + "mlp.fc2.weight"), inv_s, dbias=G(b + "mlp.fc2.bias"))
            dh = a["act"]      # the GELU output is dead after the weight gradient
            _gemm_dx(g16, [P(b + "mlp.fc2.weight")], "w", dh, aux_in=a["h"], act=ops.ACT_MUL_AUX)     # a["h"] = gelu'(pre-activation)
            linear_wgrad(dh, a["ln2"], G(b + "mlp.fc1.weight"), inv_s, dbias=G(b + "mlp.fc1.bias"))
            dln2 = _empty((M, C), torch.float32, dev)
            _gemm_dx(dh, [P(b + "mlp.fc1.weight")], "w", dln2)
            ops.layernorm_bwd(dln2, a["x2"], P(b + "norm2.weight"), a["mean2"], a["rstd2"], dy_scale=inv_s, dx_add=g, dx32=g,
                              dgamma=G(b + "norm2.weight"), dbeta=G(b + "norm2.bias"), dtype=dt)
            del dh, dln2
            # ---- window-attention branch ----
            ops.gather_rows_cast(g, g16, row_scale=a["sc1"], rows_per_scale=L, scale=S)
            linear_wgrad(g16, a["ao"], G(b + "attn.proj.weight"), inv_s, dbias=G(b + "attn.proj.bias"))
            dao = _empty((M, C), dt, dev)
            _gemm_dx(g16, [P(b + "attn.proj.weight")], "w", dao)
            dqkv = _empty((M, 3 * C), dt, dev)
            ops.win_attn_bwd(a["qkv"], dao, a["lse"], P(b + "attn.relative_position_bias_table").detach(), dqkv,
                             G(b + "attn.relative_position_bias_table"), B, res, heads, shift, 32 ** -0.5, dbias_scale=inv_s)
            linear_wgrad(dqkv, a["ln1"], G(b + "attn.qkv.weight"), inv_s, dbias=G(b + "attn.qkv.bias"))
            dln1 = _empty((M, C), torch.float32, dev)
            _gemm_dx(dqkv, [P(b + "attn.qkv.weight")], "w", dln1)
            ops.layernorm_bwd(dln1, a["x1"], P(b + "norm1.weight"), a["mean1"], a["rstd1"], dy_scale=inv_s, dx_add=g, dx32=g,
                              dgamma=G(b + "norm1.weight"), dbeta=G(b + "norm1.bias"), dtype=dt)
            del a, g16, dao, dqkv, dln1
    # ---- patch embedding ----
    pe = saved["pe"]
    L0 = spec.res0 * spec.res0
    if spec.patch_norm:
        gp = _empty((B * L0, spec.C0), torch.float32, dev)
        ops.layernorm_bwd(g, pe["x"], P("patch_embed.norm.weight"), pe["mean"], pe["rstd"], dx32=gp,
                          dgamma=G("patch_embed.norm.weight"), dbeta=G("patch_embed.norm.bias"), dtype=dt)
        g = gp
    g16 = _empty((B * L0, spec.C0), dt, dev)
    ops.gather_rows_cast(g, g16, scale=S)
    pw = P("patch_embed.proj.weight")
    kin = pw[0].numel()
    dw = torch.zeros((spec.C0, spec.kpad), dtype=torch.float32, device=dev)
    linear_wgrad(g16, pe["rows16"], dw, inv_s, dbias=G("patch_embed.proj.bias"))
    G("patch_embed.proj.weight").add_(dw[:, :kin].reshape(pw.shape))
    saved["pe"] = None


class SwinTowerFn(torch.autograd.Function):
    """pixels [B, 3, H, W] -> tokens [B, 49, 8 C0] fp32 (after the final LayerNorm); dp_scale [blocks, 2, B] of 0 / 1/keep or None."""

    @staticmethod
    def forward(ctx, spec, pixels, dp_scale, *params):
        runtime.remember_precision(ctx)
        needs_grad = any(ctx.needs_input_grad)
        out, ctx.saved = _swin_forward(spec, pixels, dp_scale, params, save=needs_grad)
        ctx.spec, ctx.params = spec, params
        return out

    @staticmethod
    @runtime.saved_precision
    def backward(ctx, dout):
        grads = GradArena(ctx.params)
        _swin_backward(ctx.spec, ctx.params, ctx.saved, dout, grads)
        ctx.saved = None
        return (None, None, None) + grads.result()
