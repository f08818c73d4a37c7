"""Per-kernel numerics: every libmico_hip.so entry point against a plain PyTorch fp32 evaluation of the same op on
the same (16-bit-rounded) inputs.  These call through the C-ABI (mico_amd.ops -> ctypes)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]


def rel_err(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-12)).item()


def tol(dtype, k=1.0):
    return (2e-3 if dtype == torch.float16 else 1.6e-2) * k


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 1), (1, 0)])
@pytest.mark.parametrize("M,N,K", [(300, 136, 200), (1024, 1408, 1408), (130, 24, 72), (257 * 3, 768, 592)])
def test_gemm_plain(cuda, dtype, ta, tb, M, N, K):
    from mico_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    ldm, ldn, ldk = ops.pad8(M), ops.pad8(N), ops.pad8(K)
    A = torch.randn((K, ldm) if ta else (M, ldk), device=cuda, generator=g).to(dtype)
    B = torch.randn((K, ldn) if tb else (N, ldk), device=cuda, generator=g).to(dtype)
    Af = (A[:, :M].t() if ta else A[:, :K]).float()
    Bf = (B[:, :N].t() if tb else B[:, :K]).float()
    ref = Af @ Bf.t()
    out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float32)
    ops.gemm(A, B, out, ta=ta, tb=tb, M=M, N=N, K=K, dtype=dtype)
    assert rel_err(out, ref) < 1e-5 * math.sqrt(K) + 1e-6
    out16 = torch.empty((M, N), device=cuda, dtype=dtype)
    ops.gemm(A, B, out16, ta=ta, tb=tb, M=M, N=N, K=K, dtype=dtype)
    assert rel_err(out16, ref) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(cuda, dtype):
    from mico_amd import ops
    torch.manual_seed(2)
    M, N, K = 514, 264, 328
    A = (0.5 * torch.randn(M, K, device=cuda)).to(dtype)
    W = (0.1 * torch.randn(N, K, device=cuda)).to(dtype)
    bias = torch.randn(N, device=cuda)
    acc = A.float() @ W.float().t()
    # bias + GELU with pre-activation copy
    h = torch.empty(M, N, device=cuda, dtype=dtype)
    a = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, W, a, bias=bias, aux_out=h, act=ops.ACT_GELU)
    assert rel_err(h, acc + bias) < tol(dtype)
    assert rel_err(a, F.gelu(acc + bias)) < tol(dtype)
    # GELU grad epilogue
    dh = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, W, dh, aux_in=h, act=ops.ACT_GELU_GRAD, alpha=0.5)
    hf = h.float()
    gp = 0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-0.5 * hf * hf) / math.sqrt(2 * math.pi)
    assert rel_err(dh, 0.5 * acc * gp) < tol(dtype)
    # the MLP pair: forward keeps gelu'(pre-activation) instead of the pre-activation, backward multiplies by it
    gd = torch.empty(M, N, device=cuda, dtype=dtype)
    a2 = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, W, a2, bias=bias, aux_out=gd, act=ops.ACT_GELU_SAVE_DERIV)
    pre = acc + bias
    gp32 = 0.5 * (1 + torch.erf(pre / math.sqrt(2))) + pre * torch.exp(-0.5 * pre * pre) / math.sqrt(2 * math.pi)
    assert rel_err(a2, F.gelu(pre)) < tol(dtype)
    assert rel_err(gd, gp32) < tol(dtype)
    Wt = W.t().contiguous()          # [K, N]: the dX orientation reads its weight reduction-major
    dh2 = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, Wt, dh2, tb=True, M=M, N=N, K=K, aux_in=gd, act=ops.ACT_MUL_AUX, alpha=0.5)
    assert rel_err(dh2, 0.5 * acc * gd.float()) < tol(dtype)
    # bias + per-sample scale + residual (in place, fp32)
    rows_per = 257
    rs = torch.rand((M + rows_per - 1) // rows_per, device=cuda) + 0.5
    x = torch.randn(M, N, device=cuda)
    ref = x + (acc + bias) * rs.repeat_interleave(rows_per)[:M, None]
    ops.gemm(A, W, x, bias=bias, row_scale=rs, rows_per_scale=rows_per, resid=x)
    assert rel_err(x, ref) < 1e-5 * math.sqrt(K)
    # patch rows -> token rows with positional table
    B_, npatch = 2, 257
    pos = torch.randn(npatch + 1, N, device=cuda)
    xt = torch.zeros(B_ * (npatch + 1), N, device=cuda)
    ops.gemm(A, W, xt, bias=bias, pos=pos, pos_rows=npatch + 1, remap=(npatch, 1, 1))
    ref = torch.zeros_like(xt).view(B_, npatch + 1, N)
    ref[:, 1:] = (acc + bias).view(B_, npatch, N) + pos[1:]
    assert rel_err(xt, ref.view(-1, N)) < 1e-5 * math.sqrt(K)
    # weight-gradient form with split-K accumulation: dW[N,K] += dY^T X
    dY = (0.1 * torch.randn(M, N, device=cuda)).to(dtype)
    dW = torch.ones(N, K, device=cuda)
    ops.gemm(dY, A, dW, ta=True, tb=True, accumulate=True, split_k=4, alpha=2.0)
    assert rel_err(dW, 1 + 2.0 * dY.float().t() @ A.float()) < 1e-5 * math.sqrt(M)


@pytest.mark.parametrize("variant", [12, 5, 8, 10])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_chip_filling_kernels(cuda, dtype, variant):
    """The large-problem kernels on a problem that fills the chip (ragged M, N = 1368 not a multiple of the 128 / 256-wide tiles): the 8-wave
    256x256x32 kernel (variant 12), the 256x128 two-workgroups-per-CU kernels (5: 32-deep stages, 8: 64-deep unit ring), the 8-phase
    256x256x64 kernel (10; its k-segment launch falls back to the 32-deep kernel) - plain forward / dX,
    the towers' residual-scatter epilogue (its own instantiation, ACT_RESID: frame map, per-frame scale, fp32 stream updated in place), the
    MLP's GELU pair, and a split-precision (2 k-segment) forward."""
    # (the product library routes by the problem alone and has no switch to force a kernel: the case runs in a process of its own on the probe
    # build that has it - tests/gemm_variant_case.py, `make -C mico_amd/csrc variants`)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "probes", "bin", "libmico_variants.so")
    assert os.path.exists(lib), "probe build missing: make -C mico_amd/csrc variants (python -c 'import __graft_entry__ as g; g.build()' builds it)"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gemm_variant_case.py"), str(variant), "f16" if dtype == torch.float16 else "bf16"],
                       env=dict(os.environ, MICO_HIP_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("dtype", DTYPES)
def test_persistent_gemm_matches_one_tile_kernel(cuda, dtype):
    """gemm_p8p_kernel (round 5: the persistent form of the 8-phase kernel - next tile's first half-tiles requested before the epilogue, 16-bit
    staging in the free ring slots, buffer stores) against gemm_p8_kernel on the same problems, bit for bit; in a process of its own on the
    probe build, whose variant 16 switches the persistent form off (tests/gemm_variant_case.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "probes", "bin", "libmico_variants.so")
    assert os.path.exists(lib), "probe build missing: make -C mico_amd/csrc variants"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gemm_variant_case.py"), "persistent", "f16" if dtype == torch.float16 else "bf16"],
                       env=dict(os.environ, MICO_HIP_LIB=lib), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(40000, 768, 256), (33001, 1536, 1408)])
def test_mlp_pair_tiled_aux_matches_row_major(cuda, dtype, M, N, K):
    """mico_gemm_epilogue::aux_tiled (round 5): the GELU pair writes gelu'(pre-activation) in the persistent kernel's accumulator layout and the
    dX launch's multiply reads it back from there - both outputs (gelu, and dX * gelu') bit for bit those of the row-major aux tensor, on a
    ragged M (rows of the last tile beyond M), plus: the library refuses the layout where it cannot honour it."""
    from mico_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = (0.5 * torch.randn(M, K, generator=g)).to(cuda).to(dtype)
    w1 = (0.05 * torch.randn(N, K, generator=g)).to(cuda).to(dtype)
    bias = torch.randn(N, generator=g).to(cuda)
    dy = torch.randn(M, 512, generator=g).to(cuda).to(dtype)
    w2 = (0.05 * torch.randn(512, N, generator=g)).to(cuda).to(dtype)       # dH = dy @ w2 : [M, N] over K2 = 512
    act_r, h_r = torch.empty(M, N, device=cuda, dtype=dtype), torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(x, w1, act_r, bias=bias, aux_out=h_r, act=ops.ACT_GELU_SAVE_DERIV)
    dh_r = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(dy, w2, dh_r, tb=True, M=M, N=N, K=512, aux_in=h_r, act=ops.ACT_MUL_AUX, alpha=0.5)
    h_t, tiled = ops.aux_buffer(M, N, K, dtype, cuda)
    assert tiled and h_t.numel() == ((M + 255) // 256) * 256 * N
    h_t.fill_(float("nan"))
    act_t = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(x, w1, act_t, bias=bias, aux_out=h_t, act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=True)
    assert ops._lib.lib().mico_gemm_last_kernel() == 8
    dh_t = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(dy, w2, dh_t, tb=True, M=M, N=N, K=512, aux_in=h_t, act=ops.ACT_MUL_AUX, alpha=0.5, aux_tiled=True)
    assert torch.equal(act_t.view(torch.int16), act_r.view(torch.int16))
    assert torch.equal(dh_t.view(torch.int16), dh_r.view(torch.int16))
    # the tiled image holds exactly the row-major values: tile (tm, tn) -> wave (wm, wn) -> unit (i, jj) -> lane (p, gq) -> 2 x 4 columns
    t = h_t.view(-1)[:((M + 255) // 256) * (N // 256) * 65536].view((M + 255) // 256, N // 256, 2, 4, 8, 2, 4, 16, 2, 4)   # tm tn wm wn i jj gq p half c
    rows = h_r.new_zeros(((M + 255) // 256) * 256, N)
    rows[:M] = h_r
    # row = tm 256 + (i >> 2) 128 + wm 64 + (i & 3) 16 + p ; col = tn 256 + wn 32 + jj 128 + half 16 + gq 4 + c
    r5 = rows.view((M + 255) // 256, 2, 2, 4, 16, N // 256, 2, 4, 2, 4, 4)        # tm ihi wm ilo p | tn jj wn half gq c
    want = r5.permute(0, 5, 2, 7, 1, 3, 6, 9, 4, 8, 10).reshape(t.shape[0], t.shape[1], 2, 4, 8, 2, 4, 16, 2, 4)
    live = torch.zeros_like(rows, dtype=torch.bool)
    live[:M] = True
    lv = live.view((M + 255) // 256, 2, 2, 4, 16, N // 256, 2, 4, 2, 4, 4).permute(0, 5, 2, 7, 1, 3, 6, 9, 4, 8, 10).reshape(t.shape)
    assert torch.equal(t[lv].view(torch.int16), want[lv].view(torch.int16))
    # refused where the launch is not the persistent pair: a small problem, and a buffer that is too small
    small = torch.empty(512, N, device=cuda, dtype=dtype)
    with pytest.raises(ops.MicoHipError):
        ops.gemm(x[:512], w1, small, bias=bias, aux_out=torch.empty(512, N, device=cuda, dtype=dtype), act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=True)
    with pytest.raises(ops.MicoHipError):
        ops.gemm(x, w1, act_t, bias=bias, aux_out=h_r, act=ops.ACT_GELU_SAVE_DERIV, aux_tiled=True)
    assert ops.aux_buffer(512, N, K, dtype, cuda)[1] is False and ops.aux_buffer(M, N + 8, K, dtype, cuda)[1] is False


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(40000, 768, 256), (33001, 1536, 1408)])
def test_mlp_pair_keeping_the_pre_activation(cuda, dtype, M, N, K):
    """Round 6: the MLP pair that keeps ONE tensor.  Forward = MICO_ACT_GELU with aux_out = the pre-activation copy, fc2's dX = MICO_ACT_GELU_GRAD with
    aux_in = that copy and aux_out = gelu(copy) (the operand of fc2's weight gradient).  The tiled instantiations of the persistent kernel
    (ACT_PRE_TILED / ACT_GRAD_TILED) against the row-major launches of the shared epilogue bit for bit, the tiled image against the row-major
    pre-activation, and everything against fp32 torch."""
    from mico_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + 1)
    x = (0.5 * torch.randn(M, K, generator=g)).to(cuda).to(dtype)
    w1 = (0.05 * torch.randn(N, K, generator=g)).to(cuda).to(dtype)
    bias = torch.randn(N, generator=g).to(cuda)
    dy = torch.randn(M, 512, generator=g).to(cuda).to(dtype)
    w2 = (0.05 * torch.randn(512, N, generator=g)).to(cuda).to(dtype)       # dH = dy @ w2 : [M, N] over K2 = 512
    # row-major (shared epilogue)
    act_r, h_r = torch.empty(M, N, device=cuda, dtype=dtype), torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(x, w1, act_r, bias=bias, aux_out=h_r, act=ops.ACT_GELU)
    dh_r, ga_r = torch.empty(M, N, device=cuda, dtype=dtype), torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    ops.gemm(dy, w2, dh_r, tb=True, M=M, N=N, K=512, aux_in=h_r, aux_out=ga_r, act=ops.ACT_GELU_GRAD, alpha=0.5)
    pre = x.float() @ w1.float().t() + bias
    assert rel_err(h_r, pre) < tol(dtype) and rel_err(act_r, F.gelu(pre)) < tol(dtype)
    hf = h_r.float()
    gp = 0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-0.5 * hf * hf) / math.sqrt(2 * math.pi)
    assert rel_err(ga_r, F.gelu(hf)) < tol(dtype)
    assert rel_err(dh_r, 0.5 * (dy.float() @ w2.float()) * gp) < tol(dtype)
    # the GELU output does not depend on whether (or how) the pre-activation is kept
    act_g = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(x, w1, act_g, bias=bias, act=ops.ACT_GELU)
    assert torch.equal(act_g.view(torch.int16), act_r.view(torch.int16))
    # tiled
    h_t, tiled = ops.aux_buffer(M, N, K, dtype, cuda)
    assert tiled
    h_t.fill_(float("nan"))
    act_t = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(x, w1, act_t, bias=bias, aux_out=h_t, act=ops.ACT_GELU, aux_tiled=True)
    assert ops._lib.lib().mico_gemm_last_kernel() == 8
    assert torch.equal(act_t.view(torch.int16), act_r.view(torch.int16))
    t = h_t.view(-1)[:((M + 255) // 256) * (N // 256) * 65536].view((M + 255) // 256, N // 256, 2, 4, 8, 2, 4, 16, 2, 4)   # tm tn wm wn i jj gq p half c
    rows = h_r.new_zeros(((M + 255) // 256) * 256, N)
    rows[:M] = h_r
    r5 = rows.view((M + 255) // 256, 2, 2, 4, 16, N // 256, 2, 4, 2, 4, 4)        # tm ihi wm ilo p | tn jj wn half gq c
    want = r5.permute(0, 5, 2, 7, 1, 3, 6, 9, 4, 8, 10).reshape(t.shape)
    live = torch.zeros_like(rows, dtype=torch.bool)
    live[:M] = True
    lv = live.view((M + 255) // 256, 2, 2, 4, 16, N // 256, 2, 4, 2, 4, 4).permute(0, 5, 2, 7, 1, 3, 6, 9, 4, 8, 10).reshape(t.shape)
    assert torch.equal(t[lv].view(torch.int16), want[lv].view(torch.int16))
    dh_t, ga_t = torch.empty(M, N, device=cuda, dtype=dtype), torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    ops.gemm(dy, w2, dh_t, tb=True, M=M, N=N, K=512, aux_in=h_t, aux_out=ga_t, act=ops.ACT_GELU_GRAD, alpha=0.5, aux_tiled=True)
    assert ops._lib.lib().mico_gemm_last_kernel() == 8
    assert torch.equal(dh_t.view(torch.int16), dh_r.view(torch.int16))
    assert torch.equal(ga_t.view(torch.int16), ga_r.view(torch.int16))
    # refused: the tiled GELU_GRAD launch takes no bias (its epilogue has no registers for one); a buffer that is too small
    with pytest.raises(ops.MicoHipError):
        ops.gemm(dy, w2, dh_t, tb=True, M=M, N=N, K=512, bias=bias, aux_in=h_t, aux_out=ga_t, act=ops.ACT_GELU_GRAD, aux_tiled=True)
    with pytest.raises(ops.MicoHipError):
        ops.gemm(dy, w2, dh_t, tb=True, M=M, N=N, K=512, aux_in=h_r, aux_out=ga_t, act=ops.ACT_GELU_GRAD, aux_tiled=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cols", [768, 1408, 2048])
@pytest.mark.parametrize("xdt", ["f32", "16"])
def test_layernorm(cuda, dtype, cols, xdt):
    from mico_amd import ops
    torch.manual_seed(3)
    rows = 1027
    x = torch.randn(rows, cols, device=cuda) * 2 + 0.3
    if xdt == "16":
        x = x.to(dtype)
    gmm = 1 + 0.1 * torch.randn(cols, device=cuda)
    bta = 0.1 * torch.randn(cols, device=cuda)
    xr = x.float().requires_grad_(True)
    gr, br = gmm.clone().requires_grad_(True), bta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (cols,), gr, br, 1e-6)
    y16 = torch.empty(rows, cols, device=cuda, dtype=dtype)
    y32 = torch.empty(rows, cols, device=cuda)
    mean = torch.empty(rows, device=cuda)
    rstd = torch.empty(rows, device=cuda)
    ops.layernorm_fwd(x, gmm, bta, 1e-6, out16=y16, out32=y32, mean=mean, rstd=rstd, dtype=dtype)
    assert rel_err(y32, ref) < 2e-6
    assert rel_err(y16, ref) < tol(dtype)
    dy = torch.randn(rows, cols, device=cuda)
    ref.backward(dy)
    add = torch.randn(rows, cols, device=cuda)
    dx = torch.empty(rows, cols, device=cuda)
    dx16 = torch.empty(rows, cols, device=cuda, dtype=dtype)
    dg = torch.ones(cols, device=cuda)
    db = torch.ones(cols, device=cuda)
    ops.layernorm_bwd(dy, x, gmm, mean, rstd, dy_scale=0.5, dx_add=add, dx32=dx, dx16=dx16, scale16=2.0, dgamma=dg, dbeta=db,
                      grad_scale=0.5, dtype=dtype)
    assert rel_err(dx, 0.5 * xr.grad + add) < 1e-5
    assert rel_err(dx16, 2.0 * (0.5 * xr.grad + add)) < tol(dtype)
    assert rel_err(dg, 1 + 0.25 * gr.grad) < 2e-5
    assert rel_err(db, 1 + 0.25 * br.grad) < 2e-5
    # dx16 with a hidden-state dropout's mask riding along (dx16_drop): the restated hash over the [rows, cols] index
    from oracle import mico_oracle as O
    dx16d = torch.empty(rows, cols, device=cuda, dtype=dtype)
    dxd = torch.empty(rows, cols, device=cuda)
    ops.layernorm_bwd(dy, x, gmm, mean, rstd, dy_scale=0.5, dx_add=add, dx32=dxd, dx16=dx16d, scale16=2.0, dtype=dtype, dx16_drop=(0.1, 77, 5))
    keep = O.drop_mask(77, 5, (rows, cols), 0.1).to(cuda)
    assert torch.equal(dxd, dx)                                       # the fp32 output does not see the mask
    assert torch.equal(dx16d == 0, (keep == 0) | (dx16d == 0)) and ((dx16d == 0) | (keep != 0)).all()
    assert rel_err(dx16d, 2.0 * (0.5 * xr.grad + add) * keep) < tol(dtype)
    # post-add table (frame + type embeddings)
    table = torch.randn(4, cols, device=cuda)
    y2 = torch.empty(rows, cols, device=cuda)
    ops.layernorm_fwd(x, gmm, bta, 1e-6, out32=y2, post_add=table, post_rows_per_group=7, post_groups=4, dtype=dtype)
    idx = (torch.arange(rows, device=cuda) // 7) % 4
    assert rel_err(y2, ref.detach() + table[idx]) < 2e-6


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("gather", [False, True])
def test_layernorm_normalised_rows(cuda, dtype, gather):
    """mico_ln_fwd_params::xhat16 / x_normalized (ABI 111, the activation diet's level 3): the forward leaves the gathered rows as fp16
    NORMALISED rows (x - mean) * rstd - half the bytes of the fp32 x_copy - without changing any of its other outputs; a later forward over
    that buffer re-creates the 16-bit output (last-bit differences only: the rows were rounded once more), and the backward over it equals
    the backward over the fp32 rows to the fp16 rounding of xhat.  fp16 rows under both compute types."""
    from mico_amd import ops
    torch.manual_seed(5)
    N, D, frames = 7, 1408, 40
    x = torch.randn(frames * N, D, device=cuda) * 3 + 0.5
    x[:, 5] *= 30.0                                               # an outlier channel, as ViT residual streams have
    gmm = 1 + 0.1 * torch.randn(D, device=cuda)
    bta = 0.1 * torch.randn(D, device=cuda)
    fmap = torch.tensor([f for f in range(frames) if f % 3 != 1], device=cuda, dtype=torch.int32) if gather else None
    rows = (fmap.numel() if gather else frames) * N
    kw = dict(frame_map=fmap, rows_per_frame=N) if gather else {}
    y0, y1 = (torch.empty(rows, D, device=cuda, dtype=dtype) for _ in range(2))
    m0, r0, m1, r1 = (torch.empty(rows, device=cuda) for _ in range(4))
    xc = torch.empty(rows, D, device=cuda)
    xh = torch.empty(rows, D, device=cuda, dtype=torch.float16)
    ops.layernorm_fwd(x, gmm, bta, 1e-6, out16=y0, mean=m0, rstd=r0, dtype=dtype, x_copy=xc, **kw)
    ops.layernorm_fwd(x, gmm, bta, 1e-6, out16=y1, mean=m1, rstd=r1, dtype=dtype, xhat16=xh, **kw)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)        # the other outputs do not notice
    ref_hat = (xc - m0[:, None]) * r0[:, None]
    assert ((xh.float() - ref_hat).abs() <= 2.0 ** -11 * ref_hat.abs() + 1e-7).all()       # one fp16 rounding
    # forward again from the normalised rows: no statistics, y = xhat gamma + beta
    y2 = torch.empty(rows, D, device=cuda, dtype=dtype)
    ops.layernorm_fwd(xh, gmm, bta, 1e-6, out16=y2, dtype=dtype, x_normalized=True)
    ulp = 2.0 ** (-10 if dtype == torch.float16 else -7)
    d = (y2.float() - y0.float()).abs()
    assert (d <= 1.5 * ulp * y0.float().abs() + 2.0 ** -11 * (gmm.abs()[None] * ref_hat.abs()) + 1e-6).all(), d.max().item()
    assert (y2 == y0).float().mean() > (0.6 if dtype == torch.float16 else 0.9)
    # backward over the normalised rows against the backward over the fp32 copy (same dy, statistics, frame scatter)
    dy = torch.randn(rows, D, device=cuda).to(dtype)
    g_a = torch.randn(frames * N, D, device=cuda)
    g_b = g_a.clone()
    dga, dba, dgb, dbb = (torch.zeros(D, device=cuda) for _ in range(4))
    ops.layernorm_bwd(dy, xc, gmm, m0, r0, dy_scale=0.25, dx_add=g_a, dx32=g_a, dgamma=dga, dbeta=dba, dtype=dtype, **kw)
    ops.layernorm_bwd(dy, xh, gmm, None, r0, dy_scale=0.25, dx_add=g_b, dx32=g_b, dgamma=dgb, dbeta=dbb, dtype=dtype, x_normalized=True, **kw)
    # dx: the rounding enters through two row means only; dgamma = sum over rows of dy * xhat is a sum of random signs, so the fp16 rounding of
    # xhat (relative 2^-12 rms per element) shows at ~1e-3 of its largest entry (tools/probes/ln_xhat_diag.py: the kernels agree with fp32
    # torch evaluations of both forms to 2e-7; the difference between the forms is the rounding itself)
    assert rel_err(g_b, g_a) < 1e-4 and rel_err(dgb, dga) < 5e-3 and rel_err(dbb, dba) < 1e-5      # (dbeta: the same sums, in the order of fp32 atomics)
    # refusals: a normalised input is fp16 and produces no statistics / copies
    from mico_amd._lib import MicoHipError
    with pytest.raises(MicoHipError):
        ops.layernorm_fwd(xh, gmm, bta, 1e-6, out16=y2, mean=m1, dtype=dtype, x_normalized=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("x_all", [False, True])
def test_layernorm_bwd_handover(cuda, dtype, x_all):
    """mico_layernorm_bwd's dx16_dst / dx16_frame_scale + mico_gather_rows_cast's dst_map: the LayerNorm backward of a branch that kept
    frames X writes the 16-bit operand of the next branch (frames Y, per-frame scale) for the frames both keep, a gather over Y \ X fills
    the rest - bit-identical to "update the stream, then gather all of Y" (the stochastic-depth backward, functional._tower_backward).
    x_all: the producing branch kept every frame (no frame_map)."""
    from mico_amd import ops
    torch.manual_seed(17)
    N, Bf, D, S = 5, 9, 256, 64.0
    keep_x = [True] * Bf if x_all else [True, False, True, True, False, False, True, True, False]
    keep_y = [False, True, True, False, False, True, True, True, True]
    fx = torch.tensor([f for f in range(Bf) if keep_x[f]], device=cuda, dtype=torch.int32)
    fy = torch.tensor([f for f in range(Bf) if keep_y[f]], device=cuda, dtype=torch.int32)
    sc = torch.rand(Bf, device=cuda) + 0.5
    rows = fx.numel() * N
    x = torch.randn(rows, D, device=cuda)
    dy = torch.randn(rows, D, device=cuda).to(dtype)
    gmm = 1 + 0.1 * torch.randn(D, device=cuda)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
    g0 = torch.randn(Bf * N, D, device=cuda)
    fmap = None if x_all else fx
    # reference schedule: update the stream in place, then gather the next branch's frames
    g_ref = g0.clone()
    ops.layernorm_bwd(dy, x, gmm, mean, rstd, dy_scale=0.25, dx_add=g_ref, dx32=g_ref, dtype=dtype, frame_map=fmap, rows_per_frame=N)
    want = torch.empty(fy.numel() * N, D, device=cuda, dtype=dtype)
    ops.gather_rows_cast(g_ref, want, row_scale=sc, rows_per_scale=N, scale=S, frame_map=fy, rows_per_frame=N)
    # hand-over
    pos_y = {int(f): j for j, f in enumerate(fy.tolist())}
    dst = torch.tensor([pos_y.get(int(f), -1) for f in fx.tolist()], device=cuda, dtype=torch.int32)
    only = [f for f in fy.tolist() if not keep_x[f]]
    g = g0.clone()
    got = torch.full((fy.numel() * N, D), float("nan"), device=cuda, dtype=dtype)
    ops.layernorm_bwd(dy, x, gmm, mean, rstd, dy_scale=0.25, dx_add=g, dx32=g, dx16=got, scale16=S, dtype=dtype, frame_map=fmap, rows_per_frame=N,
                      dx16_dst=dst, dx16_frame_scale=sc)
    if only:
        ops.gather_rows_cast(g, got, row_scale=sc, rows_per_scale=N, scale=S, frame_map=torch.tensor(only, device=cuda, dtype=torch.int32),
                             rows_per_frame=N, dst_map=torch.tensor([pos_y[f] for f in only], device=cuda, dtype=torch.int32))
    torch.cuda.synchronize()
    assert torch.equal(g, g_ref)
    assert torch.equal(got, want)


def _attn_ref(q, k, v, scale, mask):
    s = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    if mask is not None:
        s = s + (mask[:, None, None, :] if mask.dim() == 2 else mask[:, None])
    p = s.softmax(-1)
    return torch.einsum("bhij,bjhd->bihd", p, v)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,hd", [(65, 64), (129, 88), (197, 64), (256, 88), (257, 88), (257, 64), (260, 64), (261, 88), (272, 88)])
def test_attention_self_rowwise(cuda, dtype, S, hd):
    """Unmasked self-attention at every row-count class of the K/V-resident kernel (<= 256 rows: main pass only; 257..272: the
    17th query block split over the keys; 273: back on the tiled kernel) - checked ROW BY ROW (a global norm would hide one bad
    token among 257), together with the log-sum-exp the backward consumes."""
    from mico_amd import ops
    torch.manual_seed(S)
    B, H = 2, 3
    D = H * hd
    qkv = (0.7 * torch.randn(B, S, 3 * D, device=cuda)).to(dtype)
    qkv[:, S - 1, 2 * D:] *= 30      # a dropped last key (the ragged 17th sub-tile) must not hide in the noise
    qkv[:, 0, 2 * D:] *= 30
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    scale = hd ** -0.5
    qf, kf, vf = (t.float().reshape(B, S, H, hd) for t in (q, k, v))
    sc = torch.einsum("bihd,bjhd->bhij", qf, kf) * scale
    ref = torch.einsum("bhij,bjhd->bihd", sc.softmax(-1), vf).reshape(B, S, D)
    o = torch.full((B, S, D), float("nan"), device=cuda, dtype=dtype)
    lse = torch.full((B, H, S), float("nan"), device=cuda)
    ops.attn_fwd(q, k, v, o, lse, B=B, H=H, Sq=S, Sk=S, hd=hd, scale=scale, q_strides=(S * 3 * D, 3 * D),
                 k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
    torch.cuda.synchronize()
    row_err = (o.float() - ref).norm(dim=-1) / ref.norm(dim=-1)
    assert torch.isfinite(row_err).all() and row_err.max() < tol(dtype, 4), (row_err.max(), row_err.argmax())
    assert (lse - sc.logsumexp(-1)).abs().max() < 2e-3
    # backward, row by row as well (dQ rows 256.. are summed from eight key slices, dK / dV of key 256 likewise)
    qr, kr, vr = (t.detach().clone().requires_grad_(True) for t in (qf, kf, vf))
    refo = torch.einsum("bhij,bjhd->bihd", (torch.einsum("bihd,bjhd->bhij", qr, kr) * scale).softmax(-1), vr)
    do = torch.randn(B, S, D, device=cuda).to(dtype)
    refo.backward(do.float().reshape(B, S, H, hd))
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty(B, H, S, device=cuda)
    ops.attn_bwd(q, k, v, o, do, lse, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], delta, B=B, H=H, Sq=S, Sk=S, hd=hd,
                 scale=scale, q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D),
                 o_strides=(S * D, D))
    torch.cuda.synchronize()
    for name, got, want in (("dq", dqkv[..., :D], qr.grad), ("dk", dqkv[..., D:2 * D], kr.grad), ("dv", dqkv[..., 2 * D:], vr.grad)):
        want = want.reshape(B, S, D)
        err = (got.float() - want).norm(dim=-1) / (want.norm(dim=-1) + 1e-3 * want.norm(dim=-1).mean())
        assert torch.isfinite(err).all() and err.max() < tol(dtype, 8), (name, err.max(), err.argmax())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("drop", [None, (0.1, 1234, 7)])
def test_attention_shared_kv(cuda, dtype, drop):
    """mico_attn_params.kv_batch_mod: B = 6 batch entries attending to 4 K/V sets (entry i reads set i % 4 - the ITM triplet
    [own | negative | own] at b = 2) must equal the same launch over physically replicated K/V, forward and backward (dK / dV per
    batch entry; the caller adds the aliased ones), with and without attention-probability dropout (same counters: one launch)."""
    from mico_amd import ops
    torch.manual_seed(11)
    B, H, Sq, Sk, hd, nkv = 6, 12, 30, 150, 64, 4
    D = H * hd
    q = torch.randn(B, Sq, D, device=cuda).to(dtype)
    kv = torch.randn(nkv, Sk, 2 * D, device=cuda).to(dtype)
    kv_rep = kv[torch.arange(B, device=cuda) % nkv].contiguous()
    do = torch.randn(B, Sq, D, device=cuda).to(dtype)
    res = []
    for kvt, mod in ((kv, nkv), (kv_rep, 0)):
        k, v = kvt[..., :D], kvt[..., D:]
        st = dict(q_strides=(Sq * D, D), k_strides=(Sk * 2 * D, 2 * D), v_strides=(Sk * 2 * D, 2 * D), o_strides=(Sq * D, D))
        o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
        lse = torch.empty(B, H, Sq, device=cuda)
        kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=hd ** -0.5, drop=drop, kv_batch_mod=mod, **st)
        ops.attn_fwd(q, k, v, o, lse, **kw)
        dq = torch.empty_like(q)
        dkv = torch.empty(B, Sk, 2 * D, device=cuda, dtype=dtype)
        delta = torch.empty(B, H, Sq, device=cuda)
        ops.attn_bwd(q, k, v, o, do, lse, dq, dkv[..., :D], dkv[..., D:], delta, **kw)
        torch.cuda.synchronize()
        res.append((o, lse, dq, dkv))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cols,xdt,gather", [(1408, "f32", True), (1408, "f32", False), (1024, "16", False), (1792, "f32", False)])
def test_layernorm_fwd_mx8_equals_separate_quantisation(cuda, dtype, cols, xdt, gather):
    """mico_layernorm_fwd_mx8 (fp8 mode: the LayerNorm that feeds qkv / fc1 writes the GEMM's block-scaled fp8 operand itself): 16-bit output,
    statistics and the gathered fp32 copy equal mico_layernorm_fwd's bit for bit, and the e4m3 bytes + E8M0 scale words equal
    mico_quant_mx8 of that 16-bit output bit for bit - with the frame gather of the stochastic-depth path and without."""
    from mico_amd import ops
    torch.manual_seed(cols)
    rpf, frames = 257, 7
    x = (2.0 * torch.randn(frames * rpf, cols, device=cuda) + 0.3)
    x[5] = 0                                    # an all-equal row: LayerNorm output = beta
    if xdt == "16":
        x = x.to(dtype)
    g = 1.0 + 0.1 * torch.randn(cols, device=cuda)
    b = 0.05 * torch.randn(cols, device=cuda)
    b[128:160] = 0
    g[128:160] = 0                              # a 32-column block of exact zeros: scale byte of an all-zero block
    fmap = torch.tensor([6, 0, 3, 2], dtype=torch.int32, device=cuda) if gather else None
    rows = (4 if gather else frames) * rpf
    kw = dict(frame_map=fmap, rows_per_frame=rpf if gather else 0)
    outs = []
    for fused in (False, True):
        y = torch.full((rows, cols), float("nan"), device=cuda, dtype=dtype)
        mean, rstd = torch.empty(rows, device=cuda), torch.empty(rows, device=cuda)
        xc = torch.empty(rows, cols, device=cuda) if (gather and xdt == "f32") else None
        if fused:
            mx = ops.layernorm_fwd_mx8(x, g, b, 1e-6, out16=y, mean=mean, rstd=rstd, dtype=dtype, x_copy=xc, **kw)
        else:
            ops.layernorm_fwd(x, g, b, 1e-6, out16=y, mean=mean, rstd=rstd, dtype=dtype, x_copy=xc, **kw)
            mx = ops.quant_mx8(y)
        torch.cuda.synchronize()
        outs.append((y, mean, rstd, xc, mx.q, mx.scales))
    for a_, b_ in zip(*outs):
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Sq,Sk,hd,B,H", [(257, 257, 88, 40, 16), (200, 257, 72, 3, 5), (257, 256, 96, 2, 3), (130, 200, 88, 3, 4), (260, 129, 80, 2, 2),
                                          (257, 257, 88, 1, 1), (257, 257, 64, 5, 16), (197, 197, 64, 20, 12), (256, 256, 48, 2, 3)])
def test_attention_onepass_backward_shapes(cuda, dtype, Sq, Sk, hd, B, H):
    """attn_bwd_onepass_kernel beyond the towers' own shape: more items than CUs x 2 and fewer than CUs (persistent loop, XCD-contiguous and strided
    item orders), Sq != Sk (chunks from Sq, key steps / the rank-one key from Sk), every key mode (257 / 256 / < 256 with ragged and dead key
    blocks), head dims 72 .. 96 - dQ, dK, dV row by row against fp32 autograd, NaN-prefilled outputs (every row must be written)."""
    from mico_amd import ops
    torch.manual_seed(Sq * 7 + Sk)
    D = H * hd
    q = (0.8 * torch.randn(B, Sq, D, device=cuda)).to(dtype)
    kv = (0.8 * torch.randn(B, Sk, 2 * D, device=cuda)).to(dtype)
    k, v = kv[..., :D], kv[..., D:]
    scale = hd ** -0.5
    st = dict(q_strides=(Sq * D, D), k_strides=(Sk * 2 * D, 2 * D), v_strides=(Sk * 2 * D, 2 * D), o_strides=(Sq * D, D))
    kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=scale, **st)
    o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
    lse = torch.empty(B, H, Sq, device=cuda)
    ops.attn_fwd(q, k, v, o, lse, **kw)
    do = torch.randn(B, Sq, D, device=cuda).to(dtype)
    dq = torch.full_like(q, float("nan"))
    dkv = torch.full_like(kv, float("nan"))
    delta = torch.empty(B, H, Sq, device=cuda)
    ops.attn_bwd(q, k, v, o, do, lse, dq, dkv[..., :D], dkv[..., D:], delta, **kw)
    torch.cuda.synchronize()
    n = min(B, 4)      # (the reference on a few batch entries: first, last and two in between)
    sel = sorted(set([0, B - 1] + [B // 3, (2 * B) // 3][: max(0, n - 2)]))
    qr, kr, vr = (t[sel].float().reshape(len(sel), -1, H, hd).detach().requires_grad_(True) for t in (q, k, v))
    ref = torch.einsum("bhij,bjhd->bihd", (torch.einsum("bihd,bjhd->bhij", qr, kr) * scale).softmax(-1), vr)
    ref.backward(do[sel].float().reshape(len(sel), Sq, H, hd))
    assert torch.isfinite(dq).all() and torch.isfinite(dkv).all()
    for name, got, want in (("dq", dq[sel], qr.grad.reshape(len(sel), Sq, D)), ("dk", dkv[sel][..., :D], kr.grad.reshape(len(sel), Sk, D)),
                            ("dv", dkv[sel][..., D:], vr.grad.reshape(len(sel), Sk, D))):
        err = (got.float() - want).norm(dim=-1) / (want.norm(dim=-1) + 1e-3 * want.norm(dim=-1).mean())
        assert err.max() < tol(dtype, 8), (name, err.max(), err.argmax())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("drop", [None, (0.1, 77, 5)])
def test_attention_triplet_backward_in_two_launches(cuda, dtype, drop):
    """mico_attn_params.batch0 / dkv_accumulate: the ITM triplet [own | neg | own] (b = 3 n entries on 2 n K/V sets) differentiated in two
    launches - entries [0, 2 n) write dK / dV of the [own | neg] sets, entries [2 n, 3 n) ADD theirs onto the own sets, dropout counters of
    the one-launch forward - against one launch with per-entry dK / dV and the caller's add."""
    from mico_amd import ops
    torch.manual_seed(5)
    n, H, Sq, Sk, hd = 3, 12, 41, 150, 64
    B, D = 3 * n, H * hd
    q = torch.randn(B, Sq, D, device=cuda).to(dtype)
    kv = torch.randn(2 * n, Sk, 2 * D, device=cuda).to(dtype)
    do = torch.randn(B, Sq, D, device=cuda).to(dtype)
    st = dict(q_strides=(Sq * D, D), k_strides=(Sk * 2 * D, 2 * D), v_strides=(Sk * 2 * D, 2 * D), o_strides=(Sq * D, D))
    kw = dict(H=H, Sq=Sq, Sk=Sk, hd=hd, scale=hd ** -0.5, drop=drop, **st)
    k, v = kv[..., :D], kv[..., D:]
    o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
    lse = torch.empty(B, H, Sq, device=cuda)
    ops.attn_fwd(q, k, v, o, lse, B=B, kv_batch_mod=2 * n, **kw)
    delta = torch.empty(B, H, Sq, device=cuda)
    # reference: one launch, dK / dV per entry
    dq1 = torch.empty_like(q)
    dkv1 = torch.empty(B, Sk, 2 * D, device=cuda, dtype=dtype)
    ops.attn_bwd(q, k, v, o, do, lse, dq1, dkv1[..., :D], dkv1[..., D:], delta, B=B, kv_batch_mod=2 * n, **kw)
    want = dkv1[:2 * n].float()
    want[:n] += dkv1[2 * n:].float()
    # two launches
    dq2 = torch.full_like(q, float("nan"))
    dkv2 = torch.full((2 * n, Sk, 2 * D), float("nan"), device=cuda, dtype=dtype)
    ops.attn_bwd(q[:2 * n], k, v, o[:2 * n], do[:2 * n], lse[:2 * n], dq2[:2 * n], dkv2[..., :D], dkv2[..., D:], delta, B=2 * n, **kw)
    ops.attn_bwd(q[2 * n:], k, v, o[2 * n:], do[2 * n:], lse[2 * n:], dq2[2 * n:], dkv2[..., :D], dkv2[..., D:], delta, B=n, batch0=2 * n,
                 dkv_accumulate=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(dq1, dq2)
    assert torch.equal(dkv1[n:2 * n], dkv2[n:])                      # the negative sets: written once
    err = (dkv2[:n].float() - want[:n]).abs().max() / want[:n].abs().max()
    assert err < (4e-3 if dtype == torch.bfloat16 else 5e-4), err   # the own sets: one 16-bit rounding of the sum instead of two of the parts
    # the tiled kernels refuse the accumulate mode instead of ignoring it
    with pytest.raises(Exception):
        big = torch.randn(1, 200, D, device=cuda).to(dtype)
        ops.attn_bwd(big, big, big, big, big, torch.empty(1, H, 200, device=cuda), torch.empty_like(big), torch.empty_like(big), torch.empty_like(big),
                     torch.empty(1, H, 200, device=cuda), B=1, H=H, Sq=200, Sk=200, hd=hd, scale=1.0, q_strides=(200 * D, D), k_strides=(200 * D, D),
                     v_strides=(200 * D, D), o_strides=(200 * D, D), dkv_accumulate=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["vit_g", "vit_b", "bert_self2d", "bert_self3d", "bert_cross"])
def test_attention(cuda, dtype, case):
    from mico_amd import ops
    torch.manual_seed(4)
    if case == "vit_g":
        B, H, Sq, Sk, hd, mask = 3, 16, 257, 257, 88, None
    elif case == "vit_b":
        B, H, Sq, Sk, hd, mask = 2, 12, 197, 197, 64, None
    elif case == "bert_self2d":
        B, H, Sq, Sk, hd = 5, 12, 77, 77, 64
        keep = (torch.arange(Sk, device=cuda)[None] < torch.tensor([77, 30, 12, 50, 1], device=cuda)[:, None]).float()
        mask = (1 - keep) * -10000.0
    elif case == "bert_self3d":
        B, H, Sq, Sk, hd = 4, 12, 40, 40, 64
        keep = (torch.arange(Sk, device=cuda)[None] < torch.tensor([40, 30, 12, 7], device=cuda)[:, None]).float()
        mask = (1 - torch.tril(keep[:, None, :].expand(B, Sq, Sk))) * -10000.0
        mask = mask.contiguous()
    else:
        B, H, Sq, Sk, hd, mask = 3, 12, 77, 1285, 64, None
    scale = hd ** -0.5
    D = H * hd
    self_attn = Sq == Sk and case.startswith("vit")
    if self_attn:   # fused [B, N, 3, H, hd] projection buffer, as the ViT uses it
        qkv = torch.randn(B, Sq, 3 * D, device=cuda).to(dtype)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        strides = dict(q_strides=(Sq * 3 * D, 3 * D), k_strides=(Sk * 3 * D, 3 * D), v_strides=(Sk * 3 * D, 3 * D))
    else:
        q = torch.randn(B, Sq, D, device=cuda).to(dtype)
        k = torch.randn(B, Sk, D, device=cuda).to(dtype)
        v = torch.randn(B, Sk, D, device=cuda).to(dtype)
        strides = dict(q_strides=(Sq * D, D), k_strides=(Sk * D, D), v_strides=(Sk * D, D))
    qf = q.float().reshape(B, Sq, H, hd).detach().requires_grad_(True)
    kf = k.float().reshape(B, Sk, H, hd).detach().requires_grad_(True)
    vf = v.float().reshape(B, Sk, H, hd).detach().requires_grad_(True)
    ref = _attn_ref(qf, kf, vf, scale, mask)
    o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
    lse = torch.empty(B, H, Sq, device=cuda)
    kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=scale, mask=mask, o_strides=(Sq * D, D), **strides)
    ops.attn_fwd(q, k, v, o, lse, **kw)
    torch.cuda.synchronize()
    assert rel_err(o, ref.reshape(B, Sq, D)) < tol(dtype, 1.5)
    do = torch.randn(B, Sq, D, device=cuda).to(dtype)
    ref.backward(do.float().reshape(B, Sq, H, hd))
    if self_attn:
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:]
    else:
        dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    delta = torch.empty(B, H, Sq, device=cuda)
    ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, delta, **kw)
    torch.cuda.synchronize()
    assert rel_err(dq, qf.grad.reshape(B, Sq, D)) < tol(dtype, 3)
    assert rel_err(dk, kf.grad.reshape(B, Sk, D)) < tol(dtype, 3)
    assert rel_err(dv, vf.grad.reshape(B, Sk, D)) < tol(dtype, 3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Sq,Sk,mask_kind,drop", [(77, 1285, None, (0.1, 99, 5)), (77, 77, "2d", (0.1, 7, 3)), (40, 100, "3d", (0.25, 1, 0)),
                                                  (80, 33, None, None), (1, 65, "2d", (0.1, 5, 1)), (13, 32, None, (0.1, 5, 2))])
def test_attention_short_queries(cuda, dtype, Sq, Sk, mask_kind, drop):
    """The one-pass backward for Sq <= 80 at hd 64 (attn_bwd_smallq_kernel: BERT's self- and cross-attention) against autograd through
    the same attention with the dropout multipliers of the restated hash (oracle.drop_mask over [B, H, Sq, Sk]) - every strip class:
    whole 32-key strips, a ragged last strip, fewer strips than waves, one query, masks of both ranks; dQ / dK / dV ROW-wise."""
    from mico_amd import ops
    from oracle import mico_oracle as O
    torch.manual_seed(Sq * 1000 + Sk)
    B, H, hd = 3, 12, 64
    D = H * hd
    scale = hd ** -0.5
    mask = None
    if mask_kind == "2d":
        keep = (torch.arange(Sk, device=cuda)[None] < torch.tensor([Sk, max(1, Sk // 3), 1], device=cuda)[:, None]).float()
        mask = (1 - keep) * -10000.0
    elif mask_kind == "3d":
        keep = (torch.arange(Sk, device=cuda)[None, None] <= (torch.arange(Sq, device=cuda)[None, :, None] + Sk - Sq)).float().expand(B, Sq, Sk)
        mask = ((1 - keep) * -10000.0).contiguous()
    q = torch.randn(B, Sq, D, device=cuda).to(dtype)
    k = torch.randn(B, Sk, D, device=cuda).to(dtype)
    v = torch.randn(B, Sk, D, device=cuda).to(dtype)
    do = torch.randn(B, Sq, D, device=cuda).to(dtype)
    qf = q.float().reshape(B, Sq, H, hd).detach().requires_grad_(True)
    kf = k.float().reshape(B, Sk, H, hd).detach().requires_grad_(True)
    vf = v.float().reshape(B, Sk, H, hd).detach().requires_grad_(True)
    sc = torch.einsum("bihd,bjhd->bhij", qf, kf) * scale
    if mask is not None:
        sc = sc + (mask[:, None, None, :] if mask.dim() == 2 else mask[:, None])
    pr = sc.softmax(-1)
    if drop is not None:
        pr = pr * O.drop_mask(drop[1], drop[2], (B, H, Sq, Sk), drop[0]).to(cuda)
    ref = torch.einsum("bhij,bjhd->bihd", pr, vf)
    ref.backward(do.float().reshape(B, Sq, H, hd))
    o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
    lse = torch.empty(B, H, Sq, device=cuda)
    kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=scale, mask=mask, drop=drop, q_strides=(Sq * D, D), k_strides=(Sk * D, D), v_strides=(Sk * D, D),
              o_strides=(Sq * D, D))
    ops.attn_fwd(q, k, v, o, lse, **kw)
    dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    delta = torch.empty(B, H, Sq, device=cuda)
    ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, delta, **kw)
    torch.cuda.synchronize()
    assert rel_err(o, ref.reshape(B, Sq, D)) < tol(dtype, 1.5)
    for name, got, want in (("dq", dq, qf.grad.reshape(B, Sq, D)), ("dk", dk, kf.grad.reshape(B, Sk, D)), ("dv", dv, vf.grad.reshape(B, Sk, D))):
        assert torch.isfinite(got.float()).all(), name
        err = (got.float() - want).norm(dim=-1) / want.norm(dim=-1).max().clamp_min(1e-20)     # per row, against the largest row
        assert err.max().item() < tol(dtype, 4), (name, err.max().item(), err.argmax().item())


_FIVE_CASE = """
import sys, torch
sys.path.insert(0, sys.argv[1])
from mico_amd import ops
dtype = torch.float16 if sys.argv[3] == "f16" else torch.bfloat16
cuda = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(11)
n, H, hd, Sq, Sk = 4, 12, 64, 77, 517
B, D = 3 * n, H * hd
q = torch.randn(B, Sq, D, generator=g).to(cuda).to(dtype)
kv = torch.randn(2 * n, Sk, 2 * D, generator=g).to(cuda).to(dtype)      # [own | neg] sets, K | V per row: the ITM triplet reads them modulo 2 n
o = torch.empty(B, Sq, D, device=cuda, dtype=dtype)
lse = torch.empty(B, H, Sq, device=cuda)
ops.attn_fwd(q, kv, kv[:, :, D:], o, lse, B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=hd ** -0.5, drop=(0.1, 1234, 20), kv_batch_mod=2 * n,
             q_strides=(Sq * D, D), k_strides=(Sk * 2 * D, 2 * D), v_strides=(Sk * 2 * D, 2 * D), o_strides=(Sq * D, D))
torch.cuda.synchronize()
torch.save((o.cpu(), lse.cpu()), sys.argv[2])
"""


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_five_wave_forward_is_the_four_wave_arithmetic(cuda, dtype, tmp_path):
    """Round 6: BERT's 77 query rows in training run the tiled forward as ONE five-wave workgroup per (b, h) (attn_fwd_kernel<.., NW = 5>) instead of
    two four-wave ones of 64 + 13 rows - every K / V tile staged once.  Same per-wave arithmetic: outputs and lse bit for bit those of the four-wave
    launch (MICO_ATTN_NOFIVE=1, a process of its own: the switch is read once), on the ITM triplet's shared K/V memory with dropout."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, env in (("five", {}), ("four", {"MICO_ATTN_NOFIVE": "1"})):
        f = str(tmp_path / (name + ".pt"))
        r = subprocess.run([sys.executable, "-c", _FIVE_CASE, root, f, "f16" if dtype == torch.float16 else "bf16"], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(torch.load(f))
    (o5, l5), (o4, l4) = outs
    assert torch.isfinite(o5.float()).all() and o5.float().abs().max() > 0
    assert torch.equal(o5.view(torch.int16), o4.view(torch.int16)) and torch.equal(l5, l4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(cuda, dtype):
    from mico_amd import ops
    torch.manual_seed(5)
    # im2row == unfold, incl. zero padding and the 1-channel (audio) form
    for Cc, P in [(3, 14), (1, 14), (3, 16)]:
        px = torch.randn(3, Cc, 224, 224, device=cuda)
        kreal = Cc * P * P
        kpad = (kreal + 63) // 64 * 64
        rows = torch.full((3 * (224 // P) ** 2, kpad), float("nan"), device=cuda, dtype=dtype)
        ops.im2row(px, rows, P, kpad)
        ref = F.unfold(px, P, stride=P).transpose(1, 2).reshape(-1, kreal)
        assert torch.equal(rows[:, :kreal], ref.to(dtype))
        assert (rows[:, kreal:] == 0).all()
    # casts / gather / colsum / cls rows / add
    src = torch.randn(77, 588, device=cuda)
    dst = torch.empty(77, 640, device=cuda, dtype=dtype)
    ops.cast_f32_to_16(src, dst, cols=588, cols_pad=640, scale=2.0)
    assert torch.equal(dst[:, :588], (2.0 * src).to(dtype)) and (dst[:, 588:] == 0).all()
    back = torch.ones(77, 640, device=cuda)
    ops.cast_16_to_f32(dst, back, scale=0.5, accumulate=True)
    assert rel_err(back, 1 + 0.5 * dst.float()) < 1e-6
    g = torch.randn(2 * 257, 64, device=cuda)
    rs = torch.tensor([0.5, 2.0], device=cuda)
    out = torch.empty(2 * 256, 64, device=cuda, dtype=dtype)
    ops.gather_rows_cast(g, out, remap=(256, 1, 1), row_scale=rs, rows_per_scale=257, scale=3.0)
    ref = (g.view(2, 257, 64)[:, 1:] * rs[:, None, None] * 3.0).reshape(-1, 64)
    assert rel_err(out, ref) < tol(dtype)
    xs = torch.randn(5000, 300, device=cuda)
    cs = torch.ones(300, device=cuda)
    ops.colsum(xs, cs, scale=0.5, accumulate=True)
    assert rel_err(cs, 1 + 0.5 * xs.sum(0)) < 1e-5
    ops.colsum(xs.to(dtype), cs, accumulate=False)
    assert rel_err(cs, xs.to(dtype).float().sum(0)) < 1e-5
    x = torch.zeros(3 * 5, 32, device=cuda)
    cls, pos0 = torch.randn(32, device=cuda), torch.randn(32, device=cuda)
    ops.cls_rows(x, 3, 5, cls, pos0)
    assert torch.equal(x.view(3, 5, 32)[:, 0], (cls + pos0).expand(3, 32)) and (x.view(3, 5, 32)[:, 1:] == 0).all()
    a, b = torch.randn(1000, 64, device=cuda), torch.randn(1000, 64, device=cuda)
    y = torch.empty_like(a)
    y16 = torch.empty(1000, 64, device=cuda, dtype=dtype)
    ops.add_f32(a, b, y, y16, scale16=4.0)
    assert torch.equal(y, a + b) and rel_err(y16, 4 * (a + b)) < tol(dtype)
    # swiglu
    x1 = torch.randn(300, 2048, device=cuda).to(dtype)
    x2 = torch.randn(300, 2048, device=cuda).to(dtype)
    h = torch.empty_like(x1)
    ops.swiglu_fwd(x1, x2, h)
    x1f, x2f = x1.float().requires_grad_(True), x2.float().requires_grad_(True)
    ref = F.silu(x1f) * x2f
    assert rel_err(h, ref) < tol(dtype)
    dh = torch.randn(300, 2048, device=cuda).to(dtype)
    ref.backward(dh.float())
    d1, d2 = torch.empty_like(x1), torch.empty_like(x2)
    ops.swiglu_bwd(x1, x2, dh, d1, d2)
    assert rel_err(d1, x1f.grad) < tol(dtype) and rel_err(d2, x2f.grad) < tol(dtype)
    # rope (B, N, H, hd) tokens 1.. only, and its transpose
    B, N, H, hd = 2, 197, 12, 64
    t = torch.randn(B, N, H * hd, device=cuda).to(dtype)
    cos, sin = torch.randn(N - 1, hd, device=cuda), torch.randn(N - 1, hd, device=cuda)
    tf = t.float().view(B, N, H, hd)
    x2_ = tf[:, 1:].reshape(B, N - 1, H, hd // 2, 2)
    rot = torch.stack((-x2_[..., 1], x2_[..., 0]), -1).reshape(B, N - 1, H, hd)
    ref = torch.cat((tf[:, :1], tf[:, 1:] * cos[None, :, None] + rot * sin[None, :, None]), 1).reshape(B, N, H * hd)
    t2 = t.clone()
    ops.rope(t2, N * H * hd, H * hd, B, N, H, hd, cos, sin)
    assert rel_err(t2, ref) < tol(dtype)
    # inverse == autograd transpose
    tg = t.float().view(B, N, H, hd).clone().requires_grad_(True)
    xx = tg[:, 1:].reshape(B, N - 1, H, hd // 2, 2)
    rr = torch.stack((-xx[..., 1], xx[..., 0]), -1).reshape(B, N - 1, H, hd)
    yy = torch.cat((tg[:, :1], tg[:, 1:] * cos[None, :, None] + rr * sin[None, :, None]), 1)
    gy = torch.randn(B, N, H * hd, device=cuda).to(dtype)
    yy.backward(gy.float().view(B, N, H, hd))
    g2 = gy.clone()
    ops.rope(g2, N * H * hd, H * hd, B, N, H, hd, cos, sin, inverse=True)
    assert rel_err(g2, tg.grad.reshape(B, N, H * hd)) < tol(dtype)


def test_embed_loss_l2(cuda):
    from mico_amd import ops
    torch.manual_seed(6)
    vocab, S, b, D = 1000, 20, 6, 768
    ids = torch.randint(0, vocab, (b, S), device=cuda)
    word, pos, typ = torch.randn(vocab, D, device=cuda), torch.randn(512, D, device=cuda), torch.randn(2, D, device=cuda)
    out = torch.empty(b * S, D, device=cuda)
    ops.bert_embed_fwd(ids, word, pos, typ[0], out, S)
    assert torch.equal(out.view(b, S, D), word[ids] + typ[0] + pos[:S])
    dsum = torch.randn(b * S, D, device=cuda)
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros(D, device=cuda)
    ops.embed_scatter_add(ids, dsum, dword, dpos, dtyp, S, scale=0.5)
    ref = torch.zeros_like(word).index_add_(0, ids.view(-1), 0.5 * dsum)
    assert rel_err(dword, ref) < 1e-5
    assert rel_err(dpos[:S], 0.5 * dsum.view(b, S, D).sum(0)) < 1e-5
    assert rel_err(dtyp, 0.5 * dsum.sum(0)) < 1e-5
    # cross-entropy with label smoothing / ignore index / logits scale, fp32 and 16-bit logits
    for dt, cols, ls in [(torch.float32, 128, 0.1), (torch.float16, 30522, 0.0), (torch.bfloat16, 2, 0.0)]:
        rows = 37
        ld = (cols + 7) // 8 * 8
        logits = torch.randn(rows, ld, device=cuda).to(dt)
        tgt = torch.randint(0, cols, (rows,), device=cuda)
        tgt[::5] = -100
        lf = logits[:, :cols].float().detach().requires_grad_(True)
        ref = F.cross_entropy(lf / 0.07, tgt, label_smoothing=ls, reduction="none")
        row_loss = torch.empty(rows, device=cuda)
        dl = torch.zeros_like(logits)
        gscale = torch.tensor([2.0], device=cuda)
        ops.ce_fwd_bwd(logits, tgt, cols=cols, label_smoothing=ls, logits_scale=1 / 0.07, row_loss=row_loss, dlogits=dl,
                       dscale_ptr=gscale, dscale=0.25)
        assert rel_err(row_loss, ref) < (1e-5 if dt == torch.float32 else 1e-4)
        (0.5 * ref.sum()).backward()
        assert rel_err(dl[:, :cols], lf.grad) < (1e-5 if dt == torch.float32 else 1.6e-2)
    x = torch.randn(64, 512, device=cuda)
    y, inv = torch.empty_like(x), torch.empty(64, device=cuda)
    ops.l2norm_fwd(x, y, inv)
    xr = x.clone().requires_grad_(True)
    ref = F.normalize(xr, dim=-1)
    assert rel_err(y, ref) < 1e-6
    dy = torch.randn_like(x)
    ref.backward(dy)
    dx = torch.empty_like(x)
    ops.l2norm_bwd(dy, y, inv, dx)
    assert rel_err(dx, xr.grad) < 1e-5


@pytest.mark.parametrize("bs,W,rank", [(64, 1, 0), (64, 8, 3), (5, 2, 1), (3, 1, 0)])
def test_itm_sample(cuda, bs, W, rank):
    """mico_itm_sample (vast.py:423-440: softmax + 1e-4, own-rank diagonal zeroed, one draw per row) against the oracle's inverse-CDF
    restatement under injected uniform numbers - bit-exact indices wherever u * total is not within rounding of a CDF edge - plus the
    two hard rules of the reference: never the own diagonal, always a valid index."""
    from mico_amd import ops
    from oracle import mico_oracle as O
    g = torch.Generator().manual_seed(11 + bs + W)
    sim = torch.randn(bs, bs * W, generator=g) / 0.07 * 0.05          # cosine similarities / temperature
    sim[torch.arange(bs), rank * bs + torch.arange(bs)] += 8.0      # the positive pair dominates its row, as in training
    for trial in range(4):
        u = torch.rand(bs, generator=g)
        if trial == 0:
            u[0], u[-1] = 0.0, 0.999999
        ref, margin = O.itm_sample(sim, rank, bs, u)
        got = ops.itm_sample(sim.to(cuda), rank * bs, u.to(cuda)).cpu()
        sure = margin > 1e-5
        assert torch.equal(got[sure], ref[sure]), (got, ref)
        assert ((got - ref).abs() <= 1).all()
        assert (got != rank * bs + torch.arange(bs)).all() and (got >= 0).all() and (got < bs * W).all()
        assert bs < 32 or sure.float().mean() > 0.9


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mx8_quantise_and_gemm(cuda, dtype):
    """mico_quant_mx8 + mico_gemm_mx8 (block-scaled fp8 MFMA, BASELINE configs[4]).  (1) The quantiser: every element within half an
    e4m3 ulp of its block-scaled value, the scale the smallest power of two that avoids saturation.  (2) The GEMM against an fp32 matmul
    of the DEQUANTISED operands: the fp8 products are exact in fp32, so only accumulation order differs (1e-5 * sqrt(K)) - this pins
    the operand / scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (ragged M / N edges, several K-tiles, scales spanning 2^-6 .. 2^6 across
    blocks).  (3) Against the unquantised product: the stated fp8 tolerance."""
    from mico_amd import ops
    from common import rel_err
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 256 * 3 + 77, 256 * 2 + 40, 128 * 5
    blk = torch.exp2(torch.randint(-6, 7, (M, K // 32), device=cuda, generator=g).float()).repeat_interleave(32, 1)
    A = (torch.randn(M, K, device=cuda, generator=g) * blk).to(dtype)
    B = torch.randn(N, K, device=cuda, generator=g).to(dtype)
    qa, qb = ops.quant_mx8(A), ops.quant_mx8(B)
    da, db = qa.dequant(), qb.dequant()
    # (1) quantiser
    sc = torch.exp2(qa.scales.t().contiguous().view(torch.uint8).view(M, K // 128, 4).reshape(M, K // 32).float() - 127.0)
    amax = A.float().view(M, K // 32, 32).abs().amax(-1)
    assert (amax / sc <= 448.0).all() and ((amax / sc > 224.0) | (amax == 0)).all()
    ulp = torch.exp2(torch.floor(torch.log2(A.float().abs().clamp_min(1e-30) / sc.repeat_interleave(32, 1))).clamp_min(-6.0) - 3.0) * sc.repeat_interleave(32, 1)
    assert ((da - A.float()).abs() <= 0.5 * ulp * 1.0001).all()
    # (2) layout / arithmetic
    ref = da @ db.t()
    out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float32)
    ops.gemm_mx8(qa, qb, out, dtype=dtype)
    assert rel_err(out, ref) < 1e-5 * K ** 0.5
    bias = torch.randn(N, device=cuda)
    out16 = torch.empty((M, N), device=cuda, dtype=dtype)
    ops.gemm_mx8(qa, qb, out16, dtype=dtype, bias=bias)
    assert rel_err(out16, ref + bias) < (4e-3 if dtype == torch.float16 else 2e-2)
    # (3) the price of fp8: relative Frobenius error of the product of block-scaled e4m3 operands
    full = A.float() @ B.float().t()
    e = ((out - full).norm() / full.norm()).item()
    print("mx8 relative Frobenius error", e)
    assert e < 4e-2


@pytest.mark.parametrize("mode", ["res32", "stream"])
def test_attention_dkv_experiment_kernels(cuda, mode):
    """The round-3 dK / dV experiment kernels (32x32x16 MFMAs resident / LDS-DMA double-buffered halves; MICO_ATTN_DKV, read once per process):
    the g/14 tower shape - 257 tokens, head dim 88, fused QKV strides - against the default kernels' dK / dV in a child process."""
    import os
    import subprocess
    import sys
    code = """
import torch, sys
sys.path.insert(0, %r)
from mico_amd import ops
torch.manual_seed(3)
B, H, S, hd = 40, 16, 257, 88
D = H * hd
dev = torch.device('cuda:0')
qkv = torch.randn(B, S, 3 * D, device=dev).to(torch.float16)
q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
o = torch.empty(B, S, D, device=dev, dtype=torch.float16)
do = torch.randn(B, S, D, device=dev).to(torch.float16)
lse = torch.empty(B, H, S, device=dev); delta = torch.empty(B, H, S, device=dev)
st = dict(q_strides=(S * 3 * D, 3 * D), k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
kw = dict(B=B, H=H, Sq=S, Sk=S, hd=hd, scale=hd ** -0.5, **st)
ops.attn_fwd(q, k, v, o, lse, **kw)
dqkv = torch.full_like(qkv, float('nan'))
ops.attn_bwd(q, k, v, o, do, lse, dqkv[..., :D], dqkv[..., D:2 * D], dqkv[..., 2 * D:], delta, **kw)
torch.save(dqkv.float().cpu(), sys.argv[1])
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    outs = {}
    torch.cuda.empty_cache()   # the children need device memory of their own: give back what earlier (full-size) tests left cached in this process
    with tempfile.TemporaryDirectory() as td:
        for m in ("", mode):
            env = dict(os.environ)
            env.pop("MICO_ATTN_DKV", None)
            env.pop("MICO_ATTN_NOONEPASS", None)
            if m:   # the experiments are variants of the two-kernel path; the default is the one-pass kernel (round 4), which they cross-check here
                env["MICO_ATTN_DKV"] = m
                env["MICO_ATTN_NOONEPASS"] = "1"
                # (round 5: they live in the probe build only - `make -C mico_amd/csrc attnexp`; the product library has no such switch)
                lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "probes", "bin", "libmico_attnexp.so")
                assert os.path.exists(lib), "probe build missing: make -C mico_amd/csrc attnexp"
                env["MICO_HIP_LIB"] = lib
            f = os.path.join(td, f"g_{m or 'default'}.pt")
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env)
            outs[m] = torch.load(f)
    a, b = outs[""], outs[mode]
    assert torch.isfinite(b).all()
    assert ((a - b).abs().max() / a.abs().max()).item() < 2e-3


def test_token_mask_bit_exact(cuda):
    """mico_token_mask against the oracle's restatement of TokenMasker.perform_mask with the same uniform numbers: ids and labels bit-exact,
    including a row whose first draw selects nothing (second round stands), a row of pads only, and the 80 / 10 / 10 split."""
    from mico_amd import ops
    from mico_amd.model import TokenMasker
    from oracle import mico_oracle as O
    g = torch.Generator().manual_seed(11)
    b, S, R = 48, 77, 4
    ids = torch.randint(1000, 30000, (b, S), generator=g)
    ids[:, 0] = 101
    lens = torch.randint(2, S + 1, (b,), generator=g)
    ids = ids * (torch.arange(S)[None] < lens[:, None])
    ids[5, 1:] = 0                       # nothing maskable: stays unmasked whatever the draws
    um = torch.rand((R, b, S), generator=g)
    um[0, 7] = 0.99                      # row 7: the first round selects nothing, the second one stands
    um[0, 9] = 0.99
    um[1, 9] = 0.99                      # row 9: third round
    um[:, 11] = 0.99                     # rows 11, 12: every round empty -> one position is forced (the reference never returns a row
    um[:, 12] = 0.99                     # with maskable tokens and no label, general_module.py:71)
    uk, ut = torch.rand((b, S), generator=g), torch.rand((b, S), generator=g)
    ut[11, 0], ut[12, 0] = 0.0, 0.999
    ref_t, ref_l = O.token_masker_uniform(ids, 0.6, um, uk, ut)
    got_t, got_l = ops.token_mask(ids.to(cuda), 0.6, um.to(cuda), uk.to(cuda), ut.to(cuda), 103, 106, 30522)
    assert torch.equal(got_t.cpu(), ref_t) and torch.equal(got_l.cpu(), ref_l)
    assert (ref_l[5] == -100).all() and (ref_l[7] != -100).any() and (ref_l[9] != -100).any()
    assert (ref_l[11] != -100).sum() == 1 and ref_l[11, 1] != -100                  # first maskable position
    assert (ref_l[12] != -100).sum() == 1 and ref_l[12, int(lens[12]) - 1] != -100   # last maskable position
    assert ((ref_l != -100).sum(1)[(ids[:, 1:] != 0).any(1)] >= 1).all()
    sel = ref_l != -100
    assert 0.5 < sel[:, 1:].float().sum() / (ids[:, 1:] != 0).float().sum() < 0.7
    kinds = [(ref_t[sel] == 103).float().mean().item(), ((ref_t[sel] != 103) & (ref_t[sel] != ref_l[sel])).float().mean().item()]
    assert 0.7 < kinds[0] < 0.9 and 0.05 < kinds[1] < 0.16
    # the module: device tokens take the kernel (no host sync), injected uniforms reproduce the oracle
    tm = TokenMasker()
    t2, l2 = tm(ids.to(cuda), 0.6, uniforms=(um, uk, ut))
    assert torch.equal(t2.cpu(), ref_t) and torch.equal(l2.cpu(), ref_l)
    t3, l3 = tm(ids.to(cuda), 0.6)
    assert t3.is_cuda and ((l3 != -100).sum(1)[(ids[:, 1:] != 0).any(1)] >= 1).all()


def test_pool_video_kernel(cuda):
    """mico_pool_video_fwd / _bwd against torch.cat([x[:, :, 0:1], x[:, :, 1:].mean(2, keepdim=True)], dim=2) (mico.py:190-191) and its autograd
    gradient, fp32: ViT-g's 257 x 1408 frames, B/16's 197 x 768, the smallest legal frame (2 tokens), and an empty batch."""
    from mico_amd import functional as Fn
    for b, n, N, D in ((2, 3, 257, 1408), (1, 8, 197, 768), (3, 1, 2, 8), (0, 4, 257, 1408)):
        x = torch.randn((b, n, N, D), device=cuda, requires_grad=True)
        y = Fn.pool_video(x)
        ref = torch.cat([x[:, :, 0:1], x[:, :, 1:].mean(2, keepdim=True)], dim=2)
        assert y.shape == ref.shape == (b, n, 2, D)
        if b == 0:
            continue
        assert torch.equal(y[:, :, 0], ref[:, :, 0])
        assert (y - ref).abs().max().item() <= 2e-6 * ref.abs().max().item() + 1e-7
        w = torch.randn_like(ref)
        gx, = torch.autograd.grad((y * w).sum(), x)
        gr, = torch.autograd.grad((ref * w).sum(), x)
        assert (gx - gr).abs().max().item() <= 1e-6 * gr.abs().max().item()


@pytest.mark.parametrize("aligned", [True, False])
def test_dw_colfold_kernel(cuda, aligned):
    """mico_dw_colfold: dw += dwt . gamma[n] + dbt[m] beta[n], db += dbt - the LayerNorm affine folded into the weight gradient of the Linear it
    feeds (eva_vit_model.py:409-416: norm1 -> attn.qkv, norm2 -> mlp.fc1).  Against the torch expression, as views of a parameter-gradient arena
    (16-byte aligned: the vector form; off by one element: the element-wise form), and end to end: dy^T (xhat gamma + beta) from the pieces."""
    from mico_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, R = 4224, 1408, 3000
    off = 0 if aligned else 1
    arena = torch.randn(off + M * N + M + 8, device=cuda, generator=g)
    dw, db = arena[off:off + M * N].view(M, N), arena[off + M * N:off + M * N + M]
    dw0, db0 = dw.clone(), db.clone()
    gamma, beta = torch.randn(N, device=cuda, generator=g), torch.randn(N, device=cuda, generator=g)
    dy = (torch.randn(R, M, device=cuda, generator=g) * 0.1).half()
    xhat = torch.randn(R, N, device=cuda, generator=g).half()
    dwt, dbt = torch.zeros(M, N, device=cuda), torch.zeros(M, device=cuda)
    ops.gemm(dy, xhat, dwt, ta=True, tb=True, M=M, N=N, K=R, accumulate=True, alpha=0.5, split_k=0, colsum_out=dbt)
    ops.dw_colfold(dwt, dbt, gamma, beta, dw, db)
    assert torch.allclose(dw, dw0 + dwt * gamma + dbt[:, None] * beta, rtol=1e-6, atol=1e-6)
    assert torch.allclose(db, db0 + dbt, rtol=1e-6, atol=1e-6)
    y = xhat.float() * gamma + beta
    ref = 0.5 * dy.float().t() @ y
    assert rel_err(dw - dw0, ref) < 1e-4
    assert rel_err(db - db0, 0.5 * dy.float().sum(0)) < 1e-5
    dw1 = dw.clone()
    ops.dw_colfold(dwt, dbt, gamma, beta, dw, None)      # db = NULL: the caller owns the bias gradient
    assert torch.allclose(dw - dw1, dwt * gamma + dbt[:, None] * beta, rtol=1e-5, atol=1e-5) and torch.equal(db, db0 + dbt)
