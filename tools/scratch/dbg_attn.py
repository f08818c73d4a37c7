import torch, sys
sys.path.insert(0, "/root/repo")
from mico_amd import ops
torch.manual_seed(4)
cuda = torch.device("cuda:0")
for (B, H, S, hd) in ((3, 16, 257, 88), (2, 8, 257, 88), (1, 16, 257, 88), (3, 16, 256, 88), (40, 16, 257, 88)):
    for dtype in (torch.float16, torch.bfloat16):
        D = H * hd
        qkv = torch.randn(B, S, 3 * D, device=cuda).to(dtype)
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        scale = hd ** -0.5
        qf, kf, vf = (t.float().reshape(B, S, H, hd) for t in (q, k, v))
        sc = torch.einsum("bihd,bjhd->bhij", qf, kf) * scale
        ref = torch.einsum("bhij,bjhd->bihd", sc.softmax(-1), vf)
        o = torch.full((B, S, D), float("nan"), device=cuda, dtype=dtype)
        lse = torch.full((B, H, S), float("nan"), device=cuda)
        ops.attn_fwd(q, k, v, o, lse, B=B, H=H, Sq=S, Sk=S, hd=hd, scale=scale, q_strides=(S * 3 * D, 3 * D),
                     k_strides=(S * 3 * D, 3 * D), v_strides=(S * 3 * D, 3 * D), o_strides=(S * D, D))
        torch.cuda.synchronize()
        err = (o.float().reshape(B, S, H, hd) - ref).norm(dim=-1) / ref.norm(dim=-1)   # [B,S,H]
        bad = (err > 0.02) | ~torch.isfinite(err)
        print(B, H, S, hd, dtype, "max", float(err.max()), "bad", int(bad.sum()), "of", bad.numel())
        if bad.any():
            idx = bad.nonzero()
            print("   bad b:", sorted(set(idx[:, 0].tolist()))[:10], "rows:", sorted(set(idx[:, 1].tolist()))[:20], "heads:", sorted(set(idx[:, 2].tolist()))[:16])
