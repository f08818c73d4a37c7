import sys, torch
sys.path.insert(0, '/root/repo')
from mico_amd import ops
torch.manual_seed(5)
cuda = torch.device("cuda:0")
N, D, frames = 7, 1408, 40
for outl in (1.0, 30.0):
  for dtype in (torch.float16, torch.bfloat16):
    x = torch.randn(frames * N, D, device=cuda) * 3 + 0.5
    x[:, 5] *= outl
    gmm = 1 + 0.1 * torch.randn(D, device=cuda); bta = 0.1 * torch.randn(D, device=cuda)
    rows = frames * N
    y0 = torch.empty(rows, D, device=cuda, dtype=dtype)
    m0, r0 = (torch.empty(rows, device=cuda) for _ in range(2))
    xc = torch.empty(rows, D, device=cuda); xh = torch.empty(rows, D, device=cuda, dtype=torch.float16)
    ops.layernorm_fwd(x, gmm, bta, 1e-6, out16=y0, mean=m0, rstd=r0, dtype=dtype, x_copy=xc, xhat16=xh)
    dy = torch.randn(rows, D, device=cuda).to(dtype)
    g_a = torch.zeros(rows, D, device=cuda); g_b = g_a.clone()
    dga, dba, dgb, dbb = (torch.zeros(D, device=cuda) for _ in range(4))
    ops.layernorm_bwd(dy, xc, gmm, m0, r0, dy_scale=0.25, dx32=g_a, dgamma=dga, dbeta=dba, dtype=dtype)
    ops.layernorm_bwd(dy, xh, gmm, None, r0, dy_scale=0.25, dx32=g_b, dgamma=dgb, dbeta=dbb, dtype=dtype, x_normalized=True)
    hat = (xc - m0[:, None]) * r0[:, None]
    ta = (dy.float() * 0.25 * hat).sum(0); tb = (dy.float() * 0.25 * xh.float()).sum(0)
    mx = ta.abs().max()
    print(outl, dtype, "kernel a vs torch a", ((dga - ta).abs().max() / mx).item(), "kernel b vs torch b", ((dgb - tb).abs().max() / mx).item(),
          "torch a vs b", ((ta - tb).abs().max() / mx).item(), "kernel a vs b", ((dga - dgb).abs().max() / mx).item(), "argmax", (dga - dgb).abs().argmax().item())
