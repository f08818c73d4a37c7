"""k-segment (split-precision) GEMM: x W_hi + x W_lo and x_hi W_hi + x_lo W_hi + x_hi W_lo in one launch reproduce the fp32
product far below fp16 rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_precision_gemm(cuda):
    from mico_amd import ops
    torch.manual_seed(0)
    M, N, K = 300, 264, 448
    x = torch.randn(M, K, device=cuda)
    w = torch.randn(N, K, device=cuda) * 0.05
    ref = x.double() @ w.double().t()
    xh = x.half()
    xl = (x - xh.float()).half()
    wh = w.half()
    wl = (w - wh.float()).half()
    scale = ref.abs().max()

    def err(o):
        return ((o.double() - ref).abs().max() / scale).item()

    out = torch.empty(M, N, device=cuda)
    ops.gemm(xh, wh, out)
    e_plain = err(out)
    w2 = torch.cat((wh, wl), 1).contiguous()
    ops.gemm(xh, w2[:, :K], out, ksegs=(K, [0, 0], [0, K]))
    e_w = err(out)
    x2 = torch.cat((xh, xl), 1).contiguous()
    ops.gemm(x2[:, :K], w2[:, :K], out, ksegs=(K, [0, K, 0], [0, 0, K]))
    e_full = err(out)
    print(f"plain {e_plain:.2e}  weight-split {e_w:.2e}  full-split {e_full:.2e}")
    assert e_w < e_plain and e_full < 2e-6 and e_plain > 5e-5
    # the weight-split result equals the fp32 product of the ROUNDED activations with the exact weights
    ref_w = xh.double() @ w.double().t()
    ops.gemm(xh, w2[:, :K], out, ksegs=(K, [0, 0], [0, K]))
    assert ((out.double() - ref_w).abs().max() / scale).item() < 2e-6


@pytest.mark.parametrize("epi", ["lean16", "gelu_pair", "resid32"])
def test_weight_split_on_the_8_phase_kernel(cuda, epi):
    """The head-split blocks' forward GEMMs (x W_hi + x W_lo: two k-segments, the activation repeated, the weight halves adjacent along K) at the
    towers' sizes run on the 8-phase kernel with a wrapping A stream (GemmArgs::a_wrap) instead of the 32-deep k-segment kernels: against
    the fp32 product of the rounded activation with the exact weight, at a ragged row count, with the three epilogues those launches carry."""
    import torch.nn.functional as F
    from mico_amd import ops, _lib
    torch.manual_seed(1)
    M, N, K = 257 * 33 + 5, (6144 if epi == "gelu_pair" else 1408), 1408
    x = torch.randn(M, K, device=cuda)
    w = torch.randn(N, K, device=cuda) * 0.03
    bias = torch.randn(N, device=cuda)
    xh, wh = x.half(), w.half()
    w2 = torch.cat((wh, (w - wh.float()).half()), 1).contiguous()
    ref = xh.double() @ w.double().t() + bias.double()
    ks = (K, [0, 0], [0, K])
    if epi == "lean16":
        out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float16)
        ops.gemm(xh, w2[:, :K], out, ksegs=ks, bias=bias)
        got, want = out.double(), ref
    elif epi == "gelu_pair":
        out = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float16)
        aux = torch.full((M, N), float("nan"), device=cuda, dtype=torch.float16)
        ops.gemm(xh, w2[:, :K], out, ksegs=ks, bias=bias, aux_out=aux, act=ops.ACT_GELU_SAVE_DERIV)
        got, want = out.double(), F.gelu(ref)
    else:   # fp32 residual stream updated in place
        res = torch.randn(M, N, device=cuda)
        want = res.double() + ref
        ops.gemm(xh, w2[:, :K], res, ksegs=ks, bias=bias, resid=res)
        got = res.double()
    assert _lib.lib().mico_gemm_last_kernel() == 8          # the 8-phase kernel took it
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err < (1e-3 if epi != "resid32" else 2e-6), err
    if epi == "resid32":   # the plain fp16 product is measurably worse (the split is doing something)
        res2 = torch.zeros(M, N, device=cuda)
        ops.gemm(xh, wh, res2, bias=bias, resid=res2)
        assert ((res2.double() - ref).abs().max() / ref.abs().max()).item() > 10 * err
