"""BERT cross-attention (ITM triplet: 192 x 12 heads, 77 queries x 1285 keys, hd 64) with and without attention-probability dropout;
MICO_ATTN_NOSMALLQ=1 in the environment times the two tiled backward kernels instead of the one-pass short-query kernel."""
import sys, torch
sys.path.insert(0, "/root/repo")
from mico_amd import ops
dev = torch.device("cuda:0")
dt = torch.float16
for B, Sk in ((192, 1285), (64, 1285), (192, 77)):
    H, Sq, hd = 12, 77, 64
    D = H * hd
    q = torch.randn(B, Sq, D, device=dev).to(dt); k = torch.randn(B, Sk, D, device=dev).to(dt); v = torch.randn(B, Sk, D, device=dev).to(dt)
    o = torch.empty(B, Sq, D, device=dev, dtype=dt); do = torch.randn(B, Sq, D, device=dev).to(dt)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.empty(B, H, Sq, device=dev); delta = torch.empty(B, H, Sq, device=dev)
    kw = dict(B=B, H=H, Sq=Sq, Sk=Sk, hd=hd, scale=hd ** -0.5, q_strides=(Sq * D, D), k_strides=(Sk * D, D), v_strides=(Sk * D, D), o_strides=(Sq * D, D))
    for drop in (None, (0.1, 1234, 3)):
        for label, fn in (("fwd", lambda: ops.attn_fwd(q, k, v, o, lse, drop=drop, **kw)), ("bwd", lambda: ops.attn_bwd(q, k, v, o, do, lse, dq, dk, dv, delta, drop=drop, **kw))):
            for _ in range(2): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} Sk={Sk}", "drop" if drop else "none", label, f"{e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
