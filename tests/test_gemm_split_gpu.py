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
