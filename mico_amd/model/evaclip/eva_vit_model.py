"""Host-side mirror of the reference's EVA vision tower: same class / parameter names and call surface
(model/evaclip/eva_vit_model.py:488-659, model/evaclip/model.py:272-314, model/evaclip/factory.py:211-360), but the
modules below are parameter containers only - the arithmetic runs in mico_amd.functional.EvaTowerFn on libmico_hip.so.
"""
import math

import torch
from torch import nn

from ... import functional as Fn
from ... import runtime

# model/evaclip/model_configs/*.json (vision_cfg) - shape contract of the towers MiCo can select (mico.py:323-349)
MODEL_CONFIGS = {
    "EVA01-CLIP-g-14": dict(embed_dim=1024, width=1408, layers=40, head_width=88, mlp_ratio=4.3637, patch_size=14,
                            drop_path_rate=0.4, rope=False, naiveswiglu=False, subln=False),
    "EVA02-CLIP-B-16": dict(embed_dim=512, width=768, layers=12, head_width=64, mlp_ratio=2.6667, patch_size=16,
                            drop_path_rate=0.0, rope=True, naiveswiglu=True, subln=True),
    "EVA02-CLIP-L-14": dict(embed_dim=768, width=1024, layers=24, head_width=64, mlp_ratio=2.6667, patch_size=14,
                            drop_path_rate=0.0, rope=True, naiveswiglu=True, subln=True),
    # the post-norm tower (eva_vit_model.py:411-413): x + drop_path(norm1(attn(x))), x + drop_path(norm2(mlp(x)))
    "EVA02-CLIP-bigE-14-plus": dict(embed_dim=1024, width=1792, layers=64, head_width=112, mlp_ratio=8.571428571428571, patch_size=14,
                                    drop_path_rate=0.0, rope=False, naiveswiglu=False, subln=False, postnorm=True),
}


def _rope_tables(hd, grid, pt_seq_len=16):
    """VisionRotaryEmbeddingFast buffers (model/evaclip/rope.py:79-117): freqs_for='lang', dim = hd // 2, ft_seq_len = grid."""
    dim = hd // 2
    freqs = 1.0 / (10000 ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    t = torch.arange(grid) / grid * pt_seq_len
    f = torch.einsum("i,f->if", t, freqs).repeat_interleave(2, dim=-1)
    fr = torch.cat((f[:, None, :].expand(grid, grid, dim), f[None, :, :].expand(grid, grid, dim)), dim=-1)
    fr = fr.reshape(grid * grid, 2 * dim)
    return fr.cos().contiguous(), fr.sin().contiguous()


class _Rope(nn.Module):
    def __init__(self, hd, grid):
        super().__init__()
        cos, sin = _rope_tables(hd, grid)
        self.register_buffer("freqs_cos", cos)
        self.register_buffer("freqs_sin", sin)


class _Attention(nn.Module):
    def __init__(self, dim, subln):
        super().__init__()
        if subln:
            self.q_proj = nn.Linear(dim, dim, bias=False)
            self.k_proj = nn.Linear(dim, dim, bias=False)
            self.v_proj = nn.Linear(dim, dim, bias=False)
            self.inner_attn_ln = nn.LayerNorm(dim, eps=1e-6)
        else:
            self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden, swiglu):
        super().__init__()
        if swiglu:
            self.w1 = nn.Linear(dim, hidden)
            self.w2 = nn.Linear(dim, hidden)
            self.ffn_ln = nn.LayerNorm(hidden, eps=1e-6)
            self.w3 = nn.Linear(hidden, dim)
        else:
            self.fc1 = nn.Linear(dim, hidden)
            self.fc2 = nn.Linear(hidden, dim)


class _Block(nn.Module):
    def __init__(self, dim, hidden, subln, swiglu, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, subln)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, hidden, swiglu)
        self.drop_path_prob = drop_path


class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, embed_dim):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)


class EVAVisionTransformer(nn.Module):
    """Parameter layout and forward surface of eva_vit_model.py:488-659 (use_mean_pooling=False, no rel-pos bias, no layer
    scale - the only configuration the MiCo JSON configs select)."""

    def __init__(self, img_size=224, patch_size=16, num_classes=512, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, drop_path_rate=0.0, rope=False, naiveswiglu=False, subln=False, postnorm=False):
        super().__init__()
        assert not (postnorm and (naiveswiglu or subln or rope)), "the post-norm block is built for the plain-MLP tower (EVA02-CLIP-bigE-14-plus)"
        self.postnorm = bool(postnorm)
        self.image_size = img_size
        self.num_features = self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.naiveswiglu, self.subln = naiveswiglu, subln
        self.patch_embed = _PatchEmbed(img_size, patch_size, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.rope = _Rope(embed_dim // num_heads, img_size // patch_size) if rope else None
        hidden = int(embed_dim * mlp_ratio)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]   # eva_vit_model.py:533
        self.blocks = nn.ModuleList([_Block(embed_dim, hidden, subln, naiveswiglu, dpr[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        self.mlp_hidden = hidden
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.trunc_normal_(self.cls_token, std=0.02)
        self.apply(self._init_weights)
        for i, blk in enumerate(self.blocks):   # fix_init_weight, eva_vit_model.py:565-574
            blk.attn.proj.weight.data.div_(math.sqrt(2.0 * (i + 1)))
            (blk.mlp.w3 if naiveswiglu else blk.mlp.fc2).weight.data.div_(math.sqrt(2.0 * (i + 1)))
        self._spec = None

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # ---- engine glue ----
    def _tower_spec(self):
        named = [(n, p) for n, p in self.named_parameters() if not n.startswith("head.")]
        names = [n for n, _ in named]
        if self._spec is None or self._spec.names != names or self._spec.arch["depth_built"] != len(self.blocks):
            arch = dict(width=self.embed_dim, heads=self.num_heads, patch=self.patch_embed.patch_size[0],
                        mlp_hidden=self.mlp_hidden, rope=self.rope is not None, subln=self.subln, swiglu=self.naiveswiglu,
                        depth_built=len(self.blocks), postnorm=self.postnorm)
            rope = (self.rope.freqs_cos.float().contiguous(), self.rope.freqs_sin.float().contiguous()) if self.rope is not None else None
            self._spec = Fn.TowerSpec(arch, names, self.patch_embed.patch_shape[0], rope)
        elif self.rope is not None and self._spec.rope[0].device != self.rope.freqs_cos.device:
            self._spec.rope = (self.rope.freqs_cos.float().contiguous(), self.rope.freqs_sin.float().contiguous())
        return self._spec, [p for _, p in named]

    def _drop_path_scale(self, n_frames, device):
        """Per-sample stochastic-depth multipliers (eva_vit_model.py:121-138): 0 or 1/keep per (block, branch, frame); None when
        inactive.  Drawn on the HOST (torch's CPU generator, so torch.manual_seed governs it): the tower needs the kept-frame
        lists on the host to size its launches (functional.DropPlan) and a device draw would cost a sync per step."""
        probs = [b.drop_path_prob for b in self.blocks]
        if not self.training or max(probs) == 0.0:
            return None
        keep = 1.0 - torch.tensor(probs, dtype=torch.float32).view(-1, 1, 1)
        mask = torch.bernoulli(keep.expand(len(probs), 2, n_frames))
        return (mask / keep).contiguous()

    def forward_groups(self, groups, drop_path_scale=None):
        """groups: list of [B_g, C, H, W] pixel tensors (C = 3, or 1 for spectrograms evaluated with channel-summed patch
        weights - identical to repeating the channel 3x as mico.py:140 does).  Returns [sum B_g, N, D] fp32 tokens after
        the final LayerNorm."""
        for g in groups:
            H, W = g.shape[-2:]
            assert H == self.patch_embed.img_size[0] and W == self.patch_embed.img_size[1], \
                f"Input image size ({H}*{W}) doesn't match model ({self.patch_embed.img_size[0]}*{self.patch_embed.img_size[1]})."
        spec, params = self._tower_spec()
        if drop_path_scale is None:
            drop_path_scale = self._drop_path_scale(sum(g.shape[0] for g in groups), groups[0].device)
        return Fn.EvaTowerFn.apply(spec, tuple(groups), drop_path_scale, *params)

    def forward_features(self, x, return_all_features=False):
        out = self.forward_groups([x])
        return out if return_all_features else out[:, 0]

    def forward(self, x, return_all_features=False):
        if return_all_features:
            return self.forward_features(x, True)
        from ...functional import linear_f32
        return linear_f32(self.forward_features(x), self.head.weight, self.head.bias)


class CustomCLIP(nn.Module):
    """model/evaclip/model.py:272-314 reduced to what MiCo keeps: `.visual`, `.logit_scale`, `.encode_image`; the CLIP
    text tower is constructed-then-deleted by the reference (mico.py:419) and is not built here."""

    def __init__(self, embed_dim, vision_cfg):
        super().__init__()
        self.visual = EVAVisionTransformer(
            img_size=vision_cfg["image_size"], patch_size=vision_cfg["patch_size"], num_classes=embed_dim,
            embed_dim=vision_cfg["width"], depth=vision_cfg["layers"], num_heads=vision_cfg["width"] // vision_cfg["head_width"],
            mlp_ratio=vision_cfg["mlp_ratio"], drop_path_rate=vision_cfg["drop_path_rate"], rope=vision_cfg["rope"],
            naiveswiglu=vision_cfg["naiveswiglu"], subln=vision_cfg["subln"], postnorm=vision_cfg.get("postnorm", False))
        self.text = None
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def encode_image(self, image, normalize=False):
        f = self.visual(image)
        if normalize:
            from ...functional import l2_normalize
            f = l2_normalize(f)
        return f

    def encode_text(self, text, normalize=False):
        raise NotImplementedError("the CLIP text tower is deleted by MiCo.from_pretrained (model/mico.py:419); "
                                  "text goes through MiCo.forward_multimodal_encoder")


def create_model(model_name, force_custom_clip=True, image_size=None, layers=None, **_):
    """factory.py:211-360 for the JSON-registered EVA towers.  `layers` (not in the reference) truncates the depth for
    tests."""
    if model_name not in MODEL_CONFIGS:
        raise RuntimeError(f"Model config for {model_name} not found.")
    cfg = dict(MODEL_CONFIGS[model_name])
    embed_dim = cfg.pop("embed_dim")
    cfg["image_size"] = image_size or 224
    if layers is not None:
        cfg["layers"] = layers
    return CustomCLIP(embed_dim, cfg)
