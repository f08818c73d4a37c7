"""The large-problem GEMM path (producer/consumer kernel, 192x256 tiles; chosen when the problem has >= 128 256x256 tiles or is a
long-reduction weight gradient): every operand orientation, ragged M / N / K edges, split-K accumulation, the fused epilogues,
frame-scatter row map, k-segments - against fp32 matmul."""
import math

import pytest
import torch
import torch.nn.functional as F

from common import rel_err

pytestmark = pytest.mark.gpu


def _ops(A, B, ta, tb):
    Af = (A.t() if ta else A).float()
    Bf = (B.t() if tb else B).float()
    return Af @ Bf.t()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(4096 + 77, 2048 + 8, 512 + 24), (6000, 1408, 352), (193 * 40 + 5, 4224, 96),
                                   (5000 + 13, 1408 + 8, 704), (257 * 32, 1536, 64), (8192 + 40, 2048, 1408)])
def test_large_plain(cuda, dtype, ta, tb, M, N, K):
    from mico_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    ldm, ldn, ldk = ops.pad8(M), ops.pad8(N), ops.pad8(K)
    A = torch.randn((K, ldm) if ta else (M, ldk), device=cuda, generator=g).to(dtype)
    B = torch.randn((K, ldn) if tb else (N, ldk), device=cuda, generator=g).to(dtype)
    ref = _ops(A[:, :M] if ta else A[:, :K], B[:, :N] if tb else B[:, :K], ta, tb)
    out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float32)
    ops.gemm(A, B, out, ta=ta, tb=tb, M=M, N=N, K=K, dtype=dtype)
    assert rel_err(out, ref) < 1e-5 * math.sqrt(K) + 1e-6
    out16 = torch.empty((M, N), device=cuda, dtype=dtype)
    ops.gemm(A, B, out16, ta=ta, tb=tb, M=M, N=N, K=K, dtype=dtype)
    assert rel_err(out16, ref) < (4e-3 if dtype == torch.float16 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_large_weight_gradient_split_k(cuda, dtype):
    """dW += alpha * dy^T x with the library-chosen K split (atomics), tokens = 257 * 37 (ragged K tail)."""
    from mico_amd import ops
    torch.manual_seed(6)
    rows, n_out, n_in = 257 * 37, 1408, 2816
    dy = (0.1 * torch.randn(rows, n_out, device=cuda)).to(dtype)
    x = torch.randn(rows, n_in, device=cuda).to(dtype)
    dw = torch.randn(n_out, n_in, device=cuda)
    ref = dw + 0.25 * (dy.float().t() @ x.float())
    ops.gemm(dy, x, dw, ta=True, tb=True, M=n_out, N=n_in, K=rows, accumulate=True, alpha=0.25, split_k=0)
    assert rel_err(dw, ref) < 2e-5 * math.sqrt(rows)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_large_epilogues(cuda, dtype):
    from mico_amd import ops
    torch.manual_seed(7)
    tol = 4e-3 if dtype == torch.float16 else 2e-2
    M, N, K = 257 * 24, 6144, 264
    A = (0.5 * torch.randn(M, K, device=cuda)).to(dtype)
    W = (0.1 * torch.randn(N, K, device=cuda)).to(dtype)
    bias = torch.randn(N, device=cuda)
    acc = A.float() @ W.float().t()
    h = torch.empty(M, N, device=cuda, dtype=dtype)
    a = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, W, a, bias=bias, aux_out=h, act=ops.ACT_GELU)
    assert rel_err(h, acc + bias) < tol and rel_err(a, F.gelu(acc + bias)) < tol
    dh = torch.empty(M, N, device=cuda, dtype=dtype)
    ops.gemm(A, W, dh, aux_in=h, act=ops.ACT_GELU_GRAD, alpha=0.5)
    hf = h.float()
    gp = 0.5 * (1 + torch.erf(hf / math.sqrt(2))) + hf * torch.exp(-0.5 * hf * hf) / math.sqrt(2 * math.pi)
    assert rel_err(dh, 0.5 * acc * gp) < tol
    # residual stream update in place, per-frame scale, frame scatter (stochastic-depth compaction): 24 kept of 31 frames
    kept = torch.tensor(sorted(torch.randperm(31)[:24].tolist()), dtype=torch.int32, device=cuda)
    scale = torch.zeros(31, device=cuda)
    scale[kept.long()] = 1.0 / 0.7
    x = torch.randn(31 * 257, N, device=cuda)
    ref = x.clone()
    rows = (kept.long()[:, None] * 257 + torch.arange(257, device=cuda)[None]).reshape(-1)
    ref[rows] += (acc + bias) / 0.7
    ops.gemm(A, W, x, bias=bias, resid=x, row_scale=scale, rows_per_scale=257, row_map=kept, rows_per_map=257)
    assert rel_err(x, ref) < 1e-5 * math.sqrt(K) + 1e-6
    # split-precision k-segments (fp16 parity configuration)
    if dtype == torch.float16:
        Kp = 320
        xx = torch.randn(M, Kp, device=cuda)
        ww = 0.05 * torch.randn(N, Kp, device=cuda)
        xh, wh = xx.half(), ww.half()
        x2 = torch.cat((xh, (xx - xh.float()).half()), 1).contiguous()
        w2 = torch.cat((wh, (ww - wh.float()).half()), 1).contiguous()
        out = torch.empty(M, N, device=cuda)
        ops.gemm(x2[:, :Kp], w2[:, :Kp], out, ksegs=(Kp, [0, Kp, 0], [0, 0, Kp]))
        ref = xx.double() @ ww.double().t()
        assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,n_out,n_in", [(257 * 37, 1408, 2816), (257 * 64 + 3, 1408 + 8, 1408), (700, 256, 128), (257 * 40, 200, 1408)])
def test_weight_gradient_with_bias_colsum(cuda, dtype, rows, n_out, n_in):
    """mico_gemm_epilogue::colsum_out: db += alpha * sum_rows dy in the launch that computes dW += alpha * dy^T x - summed by the producer
    waves of the 192x256 kernel from the staged dy panel (large problems, incl. a ragged last K-tile, a ragged M edge and split-K) or by
    the stand-alone pass the library falls back to (small problems) - against fp32 sums."""
    from mico_amd import ops
    torch.manual_seed(8)
    dy = (0.1 * torch.randn(rows, ops.pad8(n_out), device=cuda)).to(dtype)
    x = torch.randn(rows, n_in, device=cuda).to(dtype)
    dw = torch.randn(n_out, n_in, device=cuda)
    db = torch.randn(n_out, device=cuda)
    ref_w = dw + 0.25 * (dy[:, :n_out].float().t() @ x.float())
    ref_b = db + 0.25 * dy[:, :n_out].float().sum(0)
    ops.gemm(dy, x, dw, ta=True, tb=True, M=n_out, N=n_in, K=rows, accumulate=True, alpha=0.25, split_k=0, colsum_out=db)
    assert rel_err(dw, ref_w) < 2e-5 * math.sqrt(rows)
    assert rel_err(db, ref_b) < 2e-5 * math.sqrt(rows)


@pytest.mark.parametrize("rows,n_out,n_in,split", [(257 * 37, 1408, 2816, 0), (257 * 64 + 3, 1408 + 8, 1408, 5), (4928, 768, 768, 0), (4928, 3072, 768, 7)])
def test_split_k_slabs_equal_atomics(cuda, rows, n_out, n_in, split):
    """mico_gemm_epilogue::splitk_ws: the K-splits' partial tiles through the caller's scratch + one reduction pass (both weight-gradient
    kernels: the 192x256 producer/consumer one and the 128x128 one) against the same launch without scratch (fp32 atomics) and against an
    fp32 product; a scratch too small for the requested split count must fall back, not overrun."""
    from mico_amd import ops
    torch.manual_seed(11)
    dt = torch.float16
    dy = (0.1 * torch.randn(rows, n_out, device=cuda)).to(dt)
    x = torch.randn(rows, n_in, device=cuda).to(dt)
    base = torch.randn(n_out, n_in, device=cuda)
    ref = base + 0.5 * (dy.float().t() @ x.float())
    outs = []
    for slabs in (True, False):
        ops.SPLITK_SLABS = slabs
        try:
            dw = base.clone()
            ops.gemm(dy, x, dw, ta=True, tb=True, M=n_out, N=n_in, K=rows, accumulate=True, alpha=0.5, split_k=split)
        finally:
            ops.SPLITK_SLABS = True
        assert rel_err(dw, ref) < 2e-5 * math.sqrt(rows)
        outs.append(dw)
    assert rel_err(outs[0], outs[1]) < 1e-5
    # undersized scratch: room for one slab only -> the library must not use it for a 5-way split
    e = ops._lib.lib()
    small = torch.empty(n_out * n_in, dtype=torch.float32, device=cuda)
    old = ops._splitk_scratch
    ops._splitk_scratch = lambda nbytes, device: small
    try:
        guard = torch.full((1024,), 7.0, device=cuda)
        dw = base.clone()
        ops.gemm(dy, x, dw, ta=True, tb=True, M=n_out, N=n_in, K=rows, accumulate=True, alpha=0.5, split_k=5)
        torch.cuda.synchronize()
    finally:
        ops._splitk_scratch = old
    assert rel_err(dw, ref) < 2e-5 * math.sqrt(rows) and bool((guard == 7.0).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_p8_plain_gelu_epilogue(cuda, dtype):
    """fc1 of a forward that keeps no derivative (no grad / activation diet): GELU alone on the 8-phase kernel's straight-line 16-bit
    epilogue (K % 64 == 0, no aux buffer), at a ragged row count and the towers' widths - against fp32, and equal to the value half of the
    GELU-pair launch bit for bit (same accumulation order, same GELU arithmetic)."""
    from mico_amd import ops
    torch.manual_seed(8)
    tol = 4e-3 if dtype == torch.float16 else 2e-2
    M, N, K = 257 * 40 + 3, 6144, 1408
    A = (0.5 * torch.randn(M, K, device=cuda)).to(dtype)
    W = (0.05 * torch.randn(N, K, device=cuda)).to(dtype)
    bias = torch.randn(N, device=cuda)
    ref = F.gelu(A.float() @ W.float().t() + bias)
    out = torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    ops.gemm(A, W, out, bias=bias, act=ops.ACT_GELU)
    assert rel_err(out, ref) < tol
    pair_v, pair_d = torch.empty_like(out), torch.empty_like(out)
    ops.gemm(A, W, pair_v, bias=bias, aux_out=pair_d, act=ops.ACT_GELU_SAVE_DERIV)
    assert torch.equal(out, pair_v)


def test_builtin_dma_build_agrees(cuda):
    """the default library issues the GEMM kernels' LDS-DMA by inline assembly, which hipcc's wait counters cannot see - every
    `s_waitcnt vmcnt(N)` in front of a barrier is hand-counted.  `make -C mico_amd/csrc noasm` builds the same kernels with the compiler's
    builtin (its own, conservative waits): both libraries must produce bit-identical results on the orientations / epilogues the towers
    launch - a miscounted wait shows up as a difference (an LDS race), not as a crash.  Child processes: one library per process."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(root, "tools", "probes", "bin", "libmico_noasm.so")
    if not os.path.exists(alt):
        pytest.skip("tools/probes/bin/libmico_noasm.so not built (make -C mico_amd/csrc noasm)")
    code = """
import sys, torch
sys.path.insert(0, %r)
from mico_amd import ops
dev = torch.device('cuda:0')
outs = []
for dtype in (torch.float16, torch.bfloat16):
    g = torch.Generator(device='cuda').manual_seed(3)
    M, D, Hd = 257 * 40 + 5, 1408, 6144
    x = torch.randn(M, D, device=dev, generator=g).to(dtype)
    w1 = (0.05 * torch.randn(Hd, D, device=dev, generator=g)).to(dtype)
    w2 = (0.05 * torch.randn(D, Hd, device=dev, generator=g)).to(dtype)
    a, h = torch.empty(M, Hd, device=dev, dtype=dtype), torch.empty(M, Hd, device=dev, dtype=dtype)
    ops.gemm(x, w1, a, aux_out=h, act=ops.ACT_GELU_SAVE_DERIV)                      # forward, GELU pair
    y = torch.empty(M, D, device=dev, dtype=dtype)
    ops.gemm(a, w2, y)                                                            # forward, K = 6144
    dh = torch.empty(M, Hd, device=dev, dtype=dtype)
    ops.gemm(y, w2, dh, tb=True, M=M, N=Hd, K=D, aux_in=h, act=ops.ACT_MUL_AUX)     # dX with the stored derivative
    dw = torch.zeros(Hd, D, device=dev)
    ops.gemm(dh, x, dw, ta=True, tb=True, M=Hd, N=D, K=M, accumulate=True, split_k=0)   # weight gradient (slabs: deterministic)
    outs += [a.float().cpu(), h.float().cpu(), y.float().cpu(), dh.float().cpu(), dw.cpu()]
torch.save(outs, sys.argv[1])
""" % root
    res = []
    torch.cuda.empty_cache()
    with tempfile.TemporaryDirectory() as td:
        for lib in (None, alt):
            env = dict(os.environ)
            env.pop("MICO_HIP_LIB", None)
            if lib:
                env["MICO_HIP_LIB"] = lib
            f = os.path.join(td, "o_%d.pt" % len(res))
            subprocess.run([sys.executable, "-c", code, f], check=True, env=env)
            res.append(torch.load(f))
    for i, (a, b) in enumerate(zip(*res)):
        assert torch.isfinite(a).all() and torch.equal(a, b), i


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_large_dx_fp32_accumulate_one_pass(cuda, dtype):
    """dcond += alpha * dkv W (the condition-token gradient summed over BERT's layers: fp32 C, accumulate, explicit split_k = 1, reduction-major
    B) takes the 8-phase kernel like every other large dX launch."""
    from mico_amd import ops, _lib
    torch.manual_seed(9)
    M, N, K = 64 * 1285 + 3, 768, 1536
    dy = (0.1 * torch.randn(M, K, device=cuda)).to(dtype)
    w = (0.05 * torch.randn(K, N, device=cuda)).to(dtype)
    c = torch.randn(M, N, device=cuda)
    ref = c.double() + 0.5 * (dy.double() @ w.double())
    ops.gemm(dy, w, c, tb=True, M=M, N=N, K=K, alpha=0.5, accumulate=True)
    assert _lib.lib().mico_gemm_last_kernel() == 8
    assert rel_err(c, ref) < 1e-5 * math.sqrt(K) + 1e-6
