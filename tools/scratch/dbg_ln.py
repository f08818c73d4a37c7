import torch, sys
sys.path.insert(0, "/root/repo")
from mico_amd import ops
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
for cols in (768, 1408, 2048, 64, 256):
    rows = 8
    x = torch.randn(rows, cols, device=dev) * 2 + 0.3
    g = torch.ones(cols, device=dev); b = torch.zeros(cols, device=dev)
    y32 = torch.empty(rows, cols, device=dev); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    ops.layernorm_fwd(x, g, b, 1e-6, out32=y32, mean=mean, rstd=rstd, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    print(cols, "mean", mean[:3].tolist(), "ref", x.mean(1)[:3].tolist(), "rstd", rstd[:2].tolist(), "ref", (x.var(1, unbiased=False) + 1e-6).rsqrt()[:2].tolist())
# direct: x = lane index pattern
x = torch.zeros(4, 256, device=dev)
x[0, :] = 1.0
x[1, 0] = 1.0       # lane 0 only
x[2, 4 * 17] = 1.0  # lane 17
x[3, 4 * 63] = 1.0  # lane 63
g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
y32 = torch.empty(4, 256, device=dev); mean = torch.empty(4, device=dev); rstd = torch.empty(4, device=dev)
ops.layernorm_fwd(x, g, b, 1e-6, out32=y32, mean=mean, rstd=rstd, dtype=torch.bfloat16)
print("means*256:", (mean * 256).tolist())
