"""Host-side mirror of the reference's Swin tower (model/swin.py:485-611 SwinTransformer and its sub-modules): same class, parameter and
buffer names, same constructor arguments and forward surface, so reference Swin-B / Swin-L checkpoints load by name
(data/model/general_module.py:528-578 load_swin_model).  The modules are parameter containers; the arithmetic runs in
mico_amd.swin_functional.SwinTowerFn on libmico_hip.so (window attention: csrc/swin.hip).

SURVEY.md section 8 row f4b: the reference's model/mico.py can never construct this tower (its loader lives only in the VAST sibling's
general_module.py), so it is built for the row's sake and kept off the timed path.
"""
import torch
from torch import nn

from .. import swin_functional as SF

# what general_module.py:532-539 selects: (embed_dim, depths, num_heads, vision_dim).  swin_*_patch4_window7_224_22k.yaml
SWIN_CONFIGS = {
    "swin_base_22k_224": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.2),
    "swin_large_22k_224": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], drop_path_rate=0.2),
    "swin_tiny_test": dict(embed_dim=64, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], drop_path_rate=0.1),   # test-sized (not a reference config)
}


def _relative_position_index(ws):
    """swin.py:104-114: index of the (dy, dx) offset of every token pair of a window into the (2 ws - 1)^2 bias table."""
    t = torch.arange(ws * ws)
    y, x = t // ws, t % ws
    return (y[:, None] - y[None, :] + ws - 1) * (2 * ws - 1) + (x[:, None] - x[None, :] + ws - 1)


def _shift_mask(res, ws, shift):
    """swin.py:232-253: -100 between cells of a window of the rolled grid that come from different sides of the wrap-around."""
    c = torch.arange(res)
    reg = (c >= res - ws).long() + (c >= res - shift).long()
    ids = 3 * reg[:, None] + reg[None, :]                                        # [res, res] region id of the rolled grid
    nw = res // ws
    win = ids.view(nw, ws, nw, ws).permute(0, 2, 1, 3).reshape(nw * nw, ws * ws)  # window_partition
    diff = win[:, None, :] - win[:, :, None]
    return torch.where(diff != 0, torch.full((), -100.0), torch.zeros(()))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (window_size, window_size), num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True, drop_path=0.0):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(input_resolution) <= window_size:      # swin.py:206-209
            self.shift_size, self.window_size = 0, min(input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads, qkv_bias)
        self.drop_path_prob = drop_path
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.register_buffer("attn_mask", _shift_mask(input_resolution[0], self.window_size, self.shift_size) if self.shift_size > 0 else None)


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio, qkv_bias, drop_path, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio, qkv_bias,
                                 drop_path[i]) for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, patch_norm=True):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim) if patch_norm else None


class SwinTransformer(nn.Module):
    """swin.py:485-611 with the options MiCo's loader passes (ape=False, qk_scale=None, drop / attn_drop 0, no classification head -
    the reference comments its head out and returns the normalised tokens [B, 49, num_features])."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False, fused_window_process=False, **kwargs):
        super().__init__()
        if ape or qk_scale is not None or drop_rate or attn_drop_rate or mlp_ratio != 4.0 or not qkv_bias:
            raise NotImplementedError("the MI355X Swin tower covers the configurations load_swin_model builds: ape=False, qk_scale=None, "
                                      "drop_rate=attn_drop_rate=0, mlp_ratio=4, qkv_bias=True")
        depths, num_heads = list(depths), list(num_heads)
        self.num_layers, self.embed_dim, self.patch_norm = len(depths), embed_dim, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, patch_norm)
        res = self.patch_embed.patches_resolution[0]
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths), device="cpu")]     # swin.py:541
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), (res // 2 ** i, res // 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])], i < self.num_layers - 1) for i in range(self.num_layers)])
        self.norm = nn.LayerNorm(self.num_features)
        self._cfg = dict(img_size=img_size, patch=patch_size, embed_dim=embed_dim, depths=depths, heads=num_heads, window=window_size,
                         patch_norm=patch_norm, in_chans=in_chans)
        self.apply(self._init_weights)
        self._spec = None

    @staticmethod
    def _init_weights(m):   # swin.py:571-578
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def _tower(self):
        named = list(self.named_parameters())
        names = [n for n, _ in named]
        if self._spec is None or self._spec.names != names:
            self._spec = SF.SwinSpec(names, **self._cfg)
        return self._spec, [p for _, p in named]

    def _drop_path_scale(self, n):
        """per-sample stochastic depth (swin.py:218,291,294): 0 or 1/keep per (block, branch, sample), drawn on the host"""
        probs = [blk.drop_path_prob for layer in self.layers for blk in layer.blocks]
        if not self.training or max(probs) == 0.0:
            return None
        keep = 1.0 - torch.tensor(probs, dtype=torch.float32).view(-1, 1, 1)
        return (torch.bernoulli(keep.expand(len(probs), 2, n)) / keep).contiguous()

    def forward_features(self, x, drop_path_scale=None):
        H, W = x.shape[-2:]
        assert H == self.patch_embed.img_size[0] and W == self.patch_embed.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.patch_embed.img_size[0]}*{self.patch_embed.img_size[1]})."
        spec, params = self._tower()
        if drop_path_scale is None:
            drop_path_scale = self._drop_path_scale(x.shape[0])
        if drop_path_scale is not None:
            drop_path_scale = drop_path_scale.to(x.device)
        return SF.SwinTowerFn.apply(spec, x, drop_path_scale, *params)

    def encode_audio(self, x):   # swin.py:602-606
        return self.forward_features(x.repeat(1, 3, 1, 1))

    def forward(self, x):
        return self.forward_features(x)
