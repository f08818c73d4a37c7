"""TEST INFRASTRUCTURE ONLY - the CPU oracle for the MiCo omni-modal forward/backward hot path.

A plain-PyTorch fp32 *restatement* of the reference's algorithm, written functionally over a state_dict whose
keys are the reference's own (`vision_encoder.visual.*`, `multimodal_encoder.bert.*`, ...).  It exists to check
the HIP product path (mico_amd/) - only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; mico_amd/ never does (the product path raises if the HIP library is missing, it never falls back
here).

Parity status: PINNED.  oracle/make_golden.py imports the reference itself (oracle/ref_import.py, build
container only) and writes tests/golden/*.pt; tests/test_oracle_vs_golden.py checks every function below against
those fixtures (fp32, <= 2e-5 relative).  Not pinned (third-party arithmetic absent from /root/reference, see
SURVEY.md section 8c): HF generate()/beam search, torchvision resize, torchaudio fbank.

Each function cites the reference file:line it follows (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------------------------
# architecture table: model/evaclip/model_configs/{EVA01-CLIP-g-14,EVA02-CLIP-B-16,EVA02-CLIP-L-14}.json and
# model/mico.py:323-349 (vision_encoder_type -> model name / vision_dim)
# --------------------------------------------------------------------------------------------------------------
ARCHS = {
    "evaclip01_giant": dict(width=1408, depth=40, heads=16, patch=14, mlp_hidden=6144, rope=False, subln=False,
                            swiglu=False, drop_path_rate=0.4, embed_dim=1024),
    "evaclip02_base": dict(width=768, depth=12, heads=12, patch=16, mlp_hidden=2048, rope=True, subln=True,
                           swiglu=True, drop_path_rate=0.0, embed_dim=512),
    "evaclip02_base_self": dict(width=768, depth=12, heads=12, patch=16, mlp_hidden=2048, rope=True, subln=True,
                                swiglu=True, drop_path_rate=0.0, embed_dim=512),
    "evaclip02_large": dict(width=1024, depth=24, heads=16, patch=14, mlp_hidden=2730, rope=True, subln=True,
                            swiglu=True, drop_path_rate=0.0, embed_dim=768),
    # model/evaclip/model_configs/EVA02-CLIP-bigE-14-plus.json (mico.py:341-344): the POST-norm tower (Block.forward :411-413)
    "evaclip02_bige": dict(width=1792, depth=64, heads=16, patch=14, mlp_hidden=15360, rope=False, subln=False,
                           swiglu=False, drop_path_rate=0.0, embed_dim=1024, postnorm=True),
}
VIT_EPS = 1e-6  # model/evaclip/model.py:124
BERT_EPS = 1e-12  # model/bert-base-uncased-crossattn/config.json
BERT_HEADS = 12
BERT_LAYERS = 12


def gelu(x):
    """model/mico.py:22-28 == nn.GELU (erf form)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):
    """model/evaclip/transformer.py:121-127 / torch.nn.LayerNorm."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# --------------------------------------------------------------------------------------------------------------
# EVA ViT tower
# --------------------------------------------------------------------------------------------------------------
def rope_tables(hd, grid, pt_seq_len=16):
    """model/evaclip/rope.py:79-117 (VisionRotaryEmbeddingFast, freqs_for='lang', dim = hd // 2,
    ft_seq_len = grid because intp_freq is true in the EVA02 configs)."""
    dim = hd // 2
    freqs = 1.0 / (10000 ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
    t = torch.arange(grid) / grid * pt_seq_len
    f = torch.einsum("i,f->if", t, freqs)
    f = f.repeat_interleave(2, dim=-1)  # '... n -> ... (n r)', r=2
    fh = f[:, None, :].expand(grid, grid, dim)
    fw = f[None, :, :].expand(grid, grid, dim)
    fr = torch.cat((fh, fw), dim=-1).reshape(grid * grid, 2 * dim)
    return fr.cos(), fr.sin()


def rotate_half(x):
    """model/evaclip/rope.py:23-27."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def patch_embed(sd, pre, x, patch):
    """model/evaclip/eva_vit_model.py:442-448 (conv k=s=P, flatten, transpose)."""
    y = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=patch)
    return y.flatten(2).transpose(1, 2)


def eva_attention(sd, p, x, arch, rope):
    """model/evaclip/eva_vit_model.py:293-365 (xattn is hard-wired False at :379)."""
    B, N, C = x.shape
    H = arch["heads"]
    hd = C // H
    if arch["subln"]:
        q = F.linear(x, sd[p + "q_proj.weight"], sd[p + "q_bias"])
        k = F.linear(x, sd[p + "k_proj.weight"], None)
        v = F.linear(x, sd[p + "v_proj.weight"], sd[p + "v_bias"])
        q = q.reshape(B, N, H, hd).permute(0, 2, 1, 3)
        k = k.reshape(B, N, H, hd).permute(0, 2, 1, 3)
        v = v.reshape(B, N, H, hd).permute(0, 2, 1, 3)
    else:
        qb = sd[p + "q_bias"]
        bias = torch.cat((qb, torch.zeros_like(qb), sd[p + "v_bias"]))
        qkv = F.linear(x, sd[p + "qkv.weight"], bias).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
    if rope is not None:
        cos, sin = rope
        q = torch.cat((q[:, :, :1], q[:, :, 1:] * cos + rotate_half(q[:, :, 1:]) * sin), dim=2)
        k = torch.cat((k[:, :, :1], k[:, :, 1:] * cos + rotate_half(k[:, :, 1:]) * sin), dim=2)
    q = q * hd ** -0.5
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    if arch["subln"]:
        o = layer_norm(o, sd[p + "inner_attn_ln.weight"], sd[p + "inner_attn_ln.bias"], VIT_EPS)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def eva_mlp(sd, p, x, arch):
    """model/evaclip/eva_vit_model.py:190-199 (Mlp) / :217-224 (SwiGLU)."""
    if arch["swiglu"]:
        h = F.silu(F.linear(x, sd[p + "w1.weight"], sd[p + "w1.bias"])) * F.linear(x, sd[p + "w2.weight"],
                                                                                     sd[p + "w2.bias"])
        h = layer_norm(h, sd[p + "ffn_ln.weight"], sd[p + "ffn_ln.bias"], VIT_EPS)
        return F.linear(h, sd[p + "w3.weight"], sd[p + "w3.bias"])
    h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def vit_depth(sd, pre):
    n = 0
    while (pre + f"blocks.{n}.norm1.weight") in sd:
        n += 1
    return n


def eva_vit_forward(sd, x, arch, pre="vision_encoder.visual.", drop_path_scale=None, taps=None):
    """EVAVisionTransformer.forward_features(return_all_features=True), eval mode:
    model/evaclip/eva_vit_model.py:611-650; Block.forward :409-416 (gamma None; both block orders, arch["postnorm"]).

    drop_path_scale: optional [depth, 2, B] per-sample multipliers (0 or 1/keep) standing in for the train-mode
    Bernoulli draw of drop_path (:121-138) so stochastic depth can be parity-tested with injected masks.
    taps: optional list that receives the residual stream after every block."""
    B = x.shape[0]
    t = patch_embed(sd, pre, x, arch["patch"])
    t = torch.cat((sd[pre + "cls_token"].expand(B, -1, -1), t), dim=1) + sd[pre + "pos_embed"]
    rope = None
    if arch["rope"]:
        rope = rope_tables(arch["width"] // arch["heads"], x.shape[-1] // arch["patch"])
    post = bool(arch.get("postnorm"))   # Block.forward :411-413: x + drop_path(norm1(attn(x))), x + drop_path(norm2(mlp(x)))
    for i in range(vit_depth(sd, pre)):
        p = pre + f"blocks.{i}."
        if post:
            a = layer_norm(eva_attention(sd, p + "attn.", t, arch, rope), sd[p + "norm1.weight"], sd[p + "norm1.bias"], VIT_EPS)
        else:
            a = eva_attention(sd, p + "attn.", layer_norm(t, sd[p + "norm1.weight"], sd[p + "norm1.bias"], VIT_EPS),
                              arch, rope)
        if drop_path_scale is not None:
            a = a * drop_path_scale[i, 0].view(B, 1, 1)
        t = t + a
        if post:
            m = layer_norm(eva_mlp(sd, p + "mlp.", t, arch), sd[p + "norm2.weight"], sd[p + "norm2.bias"], VIT_EPS)
        else:
            m = eva_mlp(sd, p + "mlp.", layer_norm(t, sd[p + "norm2.weight"], sd[p + "norm2.bias"], VIT_EPS), arch)
        if drop_path_scale is not None:
            m = m * drop_path_scale[i, 1].view(B, 1, 1)
        t = t + m
        if taps is not None:
            taps.append(t)
    return layer_norm(t, sd[pre + "norm.weight"], sd[pre + "norm.bias"], VIT_EPS)


# --------------------------------------------------------------------------------------------------------------
# Swin tower (model/swin.py; reachable only through data/model/general_module.py:528-578 - SURVEY.md section 8 row f4b)
# --------------------------------------------------------------------------------------------------------------
SWIN_ARCHS = {   # swin_{base,large}_patch4_window7_224_22k.yaml shape constants; swin_tiny_test is the fixture-sized variant
    "swin_base_22k_224": dict(embed_dim=128, depths=[2, 2, 18, 2], heads=[4, 8, 16, 32]),
    "swin_large_22k_224": dict(embed_dim=192, depths=[2, 2, 18, 2], heads=[6, 12, 24, 48]),
    "swin_tiny_test": dict(embed_dim=64, depths=[2, 2, 2, 2], heads=[2, 4, 8, 16]),
}
SWIN_EPS = 1e-5   # nn.LayerNorm default (swin.py:513)


def swin_window_partition(x, ws):
    """swin.py:45-57: [B, H, W, C] -> [B * nW, ws, ws, C]"""
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, C)


def swin_window_reverse(w, ws, H, W):
    """swin.py:60-74"""
    B = w.shape[0] // ((H // ws) * (W // ws))
    return w.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def swin_shift_mask(res, ws, shift):
    """swin.py:232-253: region ids painted on the rolled grid, -100 between different regions inside a window"""
    img = torch.zeros((1, res, res, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = swin_window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_rel_index(ws):
    """swin.py:104-114"""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_block(sd, p, x, res, heads, shift, ws=7, scale1=None, scale2=None):
    """SwinTransformerBlock.forward (swin.py:255-294) with WindowAttention.forward (:125-156) and Mlp (:36-42), eval mode.
    scale1 / scale2: optional per-sample [B] multipliers standing in for DropPath's train-mode draw."""
    B, L, C = x.shape
    h = layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], SWIN_EPS).view(B, res, res, C)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    xw = swin_window_partition(h, ws).view(-1, ws * ws, C)
    Bw, N = xw.shape[0], ws * ws
    qkv = F.linear(xw, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(Bw, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][swin_rel_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        mask = swin_shift_mask(res, ws, shift)
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(Bw, N, C)
    a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]).view(-1, ws, ws, C)
    a = swin_window_reverse(a, ws, res, res)
    if shift > 0:
        a = torch.roll(a, shifts=(shift, shift), dims=(1, 2))
    a = a.view(B, L, C)
    if scale1 is not None:
        a = a * scale1.view(B, 1, 1)
    x = x + a
    m = layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], SWIN_EPS)
    m = F.linear(gelu(F.linear(m, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if scale2 is not None:
        m = m * scale2.view(B, 1, 1)
    return x + m


def swin_forward(sd, x, arch, pre="vision_encoder.", drop_path_scale=None, taps=None, ws=7):
    """SwinTransformer.forward_features (swin.py:588-600): PatchEmbed (:467-475, conv 4x4 / 4 + LayerNorm), the stages
    (BasicLayer.forward :415-423: blocks, then PatchMerging :331-352), final LayerNorm.  Returns [B, 49, 8 * embed_dim]."""
    B = x.shape[0]
    t = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=4).flatten(2).transpose(1, 2)
    if (pre + "patch_embed.norm.weight") in sd:
        t = layer_norm(t, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"], SWIN_EPS)
    res = x.shape[-1] // 4
    bi = 0
    for s, (depth, heads) in enumerate(zip(arch["depths"], arch["heads"])):
        for j in range(depth):
            shift = ws // 2 if (j % 2 == 1 and res > ws) else 0       # swin.py:403, :206-209
            sc = drop_path_scale[bi] if drop_path_scale is not None else (None, None)
            t = swin_block(sd, pre + f"layers.{s}.blocks.{j}.", t, res, heads, shift, ws, sc[0], sc[1])
            bi += 1
        if taps is not None:
            taps.append(t)
        d = pre + f"layers.{s}.downsample."
        if (d + "reduction.weight") in sd:
            C = t.shape[-1]
            g = t.view(B, res, res, C)
            g = torch.cat([g[:, 0::2, 0::2], g[:, 1::2, 0::2], g[:, 0::2, 1::2], g[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
            t = F.linear(layer_norm(g, sd[d + "norm.weight"], sd[d + "norm.bias"], SWIN_EPS), sd[d + "reduction.weight"])
            res //= 2
    return layer_norm(t, sd[pre + "norm.weight"], sd[pre + "norm.bias"], SWIN_EPS)


# --------------------------------------------------------------------------------------------------------------
# BERT with cross-attention
# --------------------------------------------------------------------------------------------------------------
def extended_mask(attention_mask):
    """model/bert.py:697-781: 2-D -> [b,1,1,S], 3-D -> [b,1,S,S]; (1 - m) * -10000; no automatic causal mask."""
    m = attention_mask.to(torch.float32)
    m = m[:, None, :, :] if m.dim() == 3 else m[:, None, None, :]
    return (1.0 - m) * -10000.0


# ---- train-mode dropout (bert.py:148,267,295,373) -----------------------------------------------------------
# torch's nn.Dropout draws are not reproducible across devices, so stochastic parity is tested the same way as the
# negative-sampling and DropPath draws: with an injected mask.  The product derives its masks from a stateless
# counter hash (include/mico_hip.h, mico_dropout); drop_mask restates that hash so both sides see the same mask,
# and the rest of the arithmetic is the reference's: x * mask / (1 - p).
_DROPOUT = None          # dict(p_hidden, p_attn, seeds=iterator) while a `bert_dropout` context is active
SITE_SELF_OUT, SITE_CROSS_OUT, SITE_FFN_OUT, SITE_SELF_P, SITE_CROSS_P, SITE_EMB = 0, 1, 2, 3, 4, 100000


class bert_dropout:
    """with bert_dropout(p_hidden, p_attn, seeds): every bert_forward call inside consumes one seed (in call order)."""

    def __init__(self, p_hidden, p_attn, seeds):
        self.cfg = dict(p_hidden=p_hidden, p_attn=p_attn, seeds=iter(seeds))

    def __enter__(self):
        global _DROPOUT
        self.prev, _DROPOUT = _DROPOUT, self.cfg

    def __exit__(self, *a):
        global _DROPOUT
        _DROPOUT = self.prev


def drop_mask(seed, site, shape, p):
    """keep / (1 - p) multipliers for a contiguous tensor of `shape` (element index = flat row-major index)."""
    import numpy as np
    n = 1
    for d in shape:
        n *= d
    idx = np.arange(n, dtype=np.uint64)
    lo, hi = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = np.uint32(seed & 0xFFFFFFFF) ^ (np.uint32(site) * np.uint32(0x9E3779B9))
        h = h ^ (lo * np.uint32(0x85EBCA6B))
        h = ((h << np.uint32(13)) | (h >> np.uint32(19))) * np.uint32(5) + np.uint32(0xE6546B64)
        h = h ^ (hi * np.uint32(0xC2B2AE35))
        h = ((h << np.uint32(13)) | (h >> np.uint32(19))) * np.uint32(5) + np.uint32(0xE6546B64)
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    thr = np.uint32(int(np.float32(p) * np.float32(16777216.0)))
    keep = (h >> np.uint32(8)) >= thr
    return torch.from_numpy(keep.astype(np.float32)).view(*shape) / (1.0 - p)


def bert_attention(sd, p, hidden, kv_src, add_mask, drop=None):
    """model/bert.py:184-283 (BertSelfAttention) + :293-297 (BertSelfOutput).  drop: None or
    (p_hidden, p_attn, seed, site_probs, site_out) - the nn.Dropout calls of :267 and :295."""
    b, S, D = hidden.shape
    H, hd = BERT_HEADS, D // BERT_HEADS

    def heads(t):
        return t.view(t.shape[0], t.shape[1], H, hd).permute(0, 2, 1, 3)

    q = heads(F.linear(hidden, sd[p + "self.query.weight"], sd[p + "self.query.bias"]))
    k = heads(F.linear(kv_src, sd[p + "self.key.weight"], sd[p + "self.key.bias"]))
    v = heads(F.linear(kv_src, sd[p + "self.value.weight"], sd[p + "self.value.bias"]))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if add_mask is not None:
        s = s + add_mask
    probs = s.softmax(dim=-1)
    if drop is not None and drop[1] > 0:
        probs = probs * drop_mask(drop[2], drop[3], probs.shape, drop[1])
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(b, S, D)
    out = F.linear(ctx, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
    if drop is not None and drop[0] > 0:
        out = out * drop_mask(drop[2], drop[4], out.shape, drop[0])
    return layer_norm(out + hidden, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], BERT_EPS)


def bert_forward(sd, input_ids, attention_mask, encoder_hidden_states=None, pre="multimodal_encoder.bert.",
                 n_layers=None):
    """BertModel.forward, eval mode: model/bert.py:785-916; embeddings :101-149; BertLayer :393-461 (self-attn,
    then cross-attn when encoder_hidden_states is given - its mask is all ones -> additive 0, :872), FFN :349-375."""
    S = input_ids.shape[1]
    e = pre + "embeddings."
    x = sd[e + "word_embeddings.weight"][input_ids] + sd[e + "token_type_embeddings.weight"][0] \
        + sd[e + "position_embeddings.weight"][:S]
    x = layer_norm(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], BERT_EPS)
    dr = None
    if _DROPOUT is not None:
        dr = (_DROPOUT["p_hidden"], _DROPOUT["p_attn"], next(_DROPOUT["seeds"]))
        if dr[0] > 0:
            x = x * drop_mask(dr[2], SITE_EMB, x.shape, dr[0])
    am = extended_mask(attention_mask)
    L = n_layers
    if L is None:
        L = 0
        while (pre + f"encoder.layer.{L}.attention.self.query.weight") in sd:
            L += 1
    for i in range(L):
        p = pre + f"encoder.layer.{i}."
        x = bert_attention(sd, p + "attention.", x, x, am, dr and dr + (i * 8 + SITE_SELF_P, i * 8 + SITE_SELF_OUT))
        if encoder_hidden_states is not None:
            x = bert_attention(sd, p + "crossattention.", x, encoder_hidden_states, None,
                               dr and dr + (i * 8 + SITE_CROSS_P, i * 8 + SITE_CROSS_OUT))
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        h = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        if dr is not None and dr[0] > 0:
            h = h * drop_mask(dr[2], i * 8 + SITE_FFN_OUT, h.shape, dr[0])
        x = layer_norm(h + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], BERT_EPS)
    return x


def bert_lm_logits(sd, seq, pre="multimodal_encoder.cls."):
    """model/bert.py:575-609 (transform dense+GELU+LN, decoder tied to the word embeddings + output bias)."""
    p = pre + "predictions."
    h = gelu(F.linear(seq, sd[p + "transform.dense.weight"], sd[p + "transform.dense.bias"]))
    h = layer_norm(h, sd[p + "transform.LayerNorm.weight"], sd[p + "transform.LayerNorm.bias"], BERT_EPS)
    return F.linear(h, sd[p + "decoder.weight"], sd[p + "bias"])


def bert_mlm(sd, input_ids, attention_mask, encoder_hidden_states=None, labels=None):
    """BertForMaskedLM.forward: model/bert.py:1047-1097 -> dict(loss, logits, sequence_output)."""
    seq = bert_forward(sd, input_ids, attention_mask, encoder_hidden_states)
    logits = bert_lm_logits(sd, seq)
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1))  # ignore_index=-100
    return dict(loss=loss, logits=logits, sequence_output=seq)


# --------------------------------------------------------------------------------------------------------------
# MiCo facade (model/mico.py)
# --------------------------------------------------------------------------------------------------------------
def forward_vision_encoder(sd, arch, pixels, **kw):
    """model/mico.py:115-137."""
    b, n = pixels.shape[:2]
    out = eva_vit_forward(sd, pixels.reshape(b * n, *pixels.shape[2:]), arch, **kw)
    return out.reshape(b, n, *out.shape[-2:])


def forward_audio_encoder(sd, arch, spec, **kw):
    """model/mico.py:139-143 (1-channel spectrogram repeated to 3 channels)."""
    return forward_vision_encoder(sd, arch, spec.unsqueeze(2).repeat(1, 1, 3, 1, 1), **kw)


def pool_for_contra(feature):
    """model/mico.py:157-182 (CLS of each frame, mean over frames)."""
    return feature[:, :, 0].mean(dim=1)


def contra_feat(sd, head, pooled):
    """Contra_head (mico.py:36-41, bias-free) or the fused nn.Linear heads (:391-394), then F.normalize
    (data/model/vast.py:221-279)."""
    if (head + ".linear.weight") in sd:
        y = F.linear(pooled, sd[head + ".linear.weight"])
    else:
        y = F.linear(pooled, sd[head + ".weight"], sd[head + ".bias"])
    return F.normalize(y, dim=-1)


def multimodal_input(sd, modality, feats, pool_video=False):
    """model/mico.py:187-243: [pool_video], Linear+LN(1e-12), + frame embedding (nearest-interpolated when n differs),
    reshape (b, n*x, 768), + type embedding."""
    b, n, x, c = feats.shape
    if pool_video:
        feats = torch.cat([feats[:, :, 0:1], feats[:, :, 1:].mean(2, keepdim=True)], dim=2)
    p = f"hidden_trans_{modality}_multimodal."
    y = layer_norm(F.linear(feats, sd[p + "0.weight"], sd[p + "0.bias"]), sd[p + "1.weight"], sd[p + "1.bias"],
                   BERT_EPS)
    fe = sd[f"{modality}_frame_embedding"]
    if n != fe.shape[1]:
        fe = F.interpolate(fe.float().permute(0, 2, 1), n, mode="nearest").permute(0, 2, 1)
    y = y + fe.unsqueeze(-2)
    y = y.reshape(b, -1, y.shape[-1])
    return y + sd[f"{modality}_type_embeddings"]


def itm_head(sd, cls_tok):
    """Match_head: model/mico.py:44-52."""
    h = gelu(F.linear(cls_tok, sd["itm_head.linear1.weight"], sd["itm_head.linear1.bias"]))
    h = layer_norm(h, sd["itm_head.layernorm.weight"], sd["itm_head.layernorm.bias"], BERT_EPS)
    return F.linear(h, sd["itm_head.linear2.weight"], sd["itm_head.linear2.bias"])


# --------------------------------------------------------------------------------------------------------------
# alignment loss (spec: data/model/vast.py:383-512 generalised with MiCo's depth heads)
# --------------------------------------------------------------------------------------------------------------
COND_MODALITY = {"v": "vision", "a": "audio", "d": "depth"}


def encode_batch(sd, arch, batch, pool_video=False, drop_path_scale=None):
    """The encoder part of batch_get (vast.py:81-314): towers, pooled features, condition tensors.
    batch keys: vision_pixels [b,n,3,h,w], audio_spectrograms [b,n,h,w], depth_pixels [b,n,3,h,w] (any subset),
    input_ids / attention_mask [b,S]."""
    out = {}
    keys = {"v": "vision_pixels", "a": "audio_spectrograms", "d": "depth_pixels"}
    for m, key in keys.items():
        if key not in batch:
            continue
        kw = {}
        if drop_path_scale is not None and m in drop_path_scale:
            kw["drop_path_scale"] = drop_path_scale[m]
        if m == "a":
            o = forward_audio_encoder(sd, arch, batch[key], **kw)
        else:
            o = forward_vision_encoder(sd, arch, batch[key], **kw)
        out["output_" + m] = o
        out["pooled_" + m] = pool_for_contra(o)
        out["condition_feats_" + m] = multimodal_input(sd, COND_MODALITY[m], o, pool_video)
    if "subtitle_ids" in batch:   # vast.py:168-174, 196-199, 236-238: subtitles go through the text BERT, CLS pooled
        sub_seq = bert_forward(sd, batch["subtitle_ids"], batch["subtitle_mask"])
        out["output_s"] = sub_seq
        out["pooled_s"] = sub_seq[:, 0]
        y = layer_norm(F.linear(sub_seq, sd["hidden_trans_subtitle_multimodal.0.weight"], sd["hidden_trans_subtitle_multimodal.0.bias"]),
                       sd["hidden_trans_subtitle_multimodal.1.weight"], sd["hidden_trans_subtitle_multimodal.1.bias"], BERT_EPS)
        out["condition_feats_s"] = y + sd["subtitle_type_embeddings"]          # mico.py:245-248
    if "input_ids" in batch:
        seq = bert_forward(sd, batch["input_ids"], batch["attention_mask"])
        out["caption_output"] = seq
        out["feat_t"] = contra_feat(sd, "contra_head_t", seq[:, 0])
    return out


FUSED_HEADS = {"v": "contra_head_v", "a": "contra_head_a", "d": "contra_head_d", "s": "contra_head_s", "va": "contra_head_va",
               "vd": "contra_head_id", "vs": "contra_head_vs", "vas": "contra_head_vas"}


def feat_cond(sd, enc, cond):
    """feat_<cond> of vast.py:221-279: single-modality Contra_head, or the fused Linear over concatenated pooled
    features (contra_head_va; MiCo adds contra_head_id for image+depth, mico.py:392)."""
    pooled = torch.cat([enc["pooled_" + m] for m in cond], dim=1)
    return contra_feat(sd, FUSED_HEADS[cond], pooled)


def condition_feats(enc, cond):
    """condition_feats_<cond>: concatenation along the token axis (vast.py:192-210)."""
    return torch.cat([enc["condition_feats_" + m] for m in cond], dim=1)


def itc_loss(feat_t, feat_c, feat_t_all, feat_c_all, temp, rank, label_smoothing=0.1):
    """vast.py:405-415.  *_all are the gathered (no-grad) global-batch features."""
    bs = feat_t.shape[0]
    sim_c2t = feat_c @ feat_t_all.t() / temp
    sim_t2c = feat_t @ feat_c_all.t() / temp
    targets = torch.arange(rank * bs, rank * bs + bs, device=feat_t.device)
    loss = (F.cross_entropy(sim_c2t, targets, label_smoothing=label_smoothing)
            + F.cross_entropy(sim_t2c, targets, label_smoothing=label_smoothing)) / 2
    return loss, sim_t2c, sim_c2t


def itm_weights(sim, rank):
    """vast.py:423-427: softmax + 1e-4 with the own-rank diagonal zeroed (no grad)."""
    bs = sim.shape[0]
    w = F.softmax(sim.detach(), dim=1) + 1e-4
    w[:, rank * bs: rank * bs + bs].fill_diagonal_(0)
    return w


def itm_loss(sd, input_ids, attention_mask, cond, cond_all, ids_all, mask_all, neg_cond_idx, neg_text_idx,
             itm_ratio):
    """vast.py:429-457 with the multinomial draws replaced by injected indices (neg_*_idx [bs], int64)."""
    bs = input_ids.shape[0]
    ids = torch.cat((input_ids, input_ids, ids_all[neg_text_idx]), dim=0)
    am = torch.cat((attention_mask, attention_mask, mask_all[neg_text_idx]), dim=0)
    c = torch.cat((cond, cond_all[neg_cond_idx], cond), dim=0)
    out = bert_forward(sd, ids, am, c)
    logits = itm_head(sd, out[:, 0])
    gt = torch.zeros(bs * 3, dtype=torch.long, device=logits.device)
    gt[:bs] = 1
    return itm_ratio * F.cross_entropy(logits, gt), logits


def cap_loss(sd, masked_ids, attention_mask, labels, cond):
    """vast.py:493-509: 3-D mask = key padding AND lower-triangular; BertForMaskedLM(..., labels).loss."""
    S = attention_mask.shape[1]
    m3 = attention_mask.unsqueeze(1).expand(-1, S, -1).clone()
    m3 = torch.tril(m3)
    return bert_mlm(sd, masked_ids, m3, cond, labels)["loss"]


def mico_forward(sd, arch, batch, task, cfg, rank=0, world=None, injected=None):
    """MiCo.forward(batch, task, compute_loss=True) as specified by VAST.forward (vast.py:317-348) /
    forward_ret (:383-464) / forward_cap (:485-512).  Single-process: `world` optionally holds the other ranks'
    gathered tensors ({'feat_t_all','ids_all','mask_all', 'feat_<c>_all', 'cond_<c>_all'}); by default the
    global batch is the local batch (W=1).  `injected` carries the RNG draws: neg_cond_idx / neg_text_idx per
    subtask, masked_ids / labels for cap."""
    enc = encode_batch(sd, arch, batch, cfg.get("pool_video", False), (injected or {}).get("drop_path_scale"))
    ids, am = batch["input_ids"], batch["attention_mask"]
    out = {}
    for t in task.split("_"):
        subtasks = t.split("%")[1:]
        if t.startswith("ret"):
            l_itc, l_itm = [], []
            feat_t = enc["feat_t"]
            feat_t_all = world["feat_t_all"] if world else feat_t.detach()
            ids_all = world["ids_all"] if world else ids
            mask_all = world["mask_all"] if world else am
            for st in subtasks:
                c = st[1:]
                fc = feat_cond(sd, enc, c)
                fc_all = world[f"feat_{c}_all"] if world else fc.detach()
                li, sim_t2c, sim_c2t = itc_loss(feat_t, fc, feat_t_all, fc_all, sd["contra_temp"], rank)
                l_itc.append(li)
                cond = condition_feats(enc, c)
                cond_all = world[f"cond_{c}_all"] if world else cond
                inj = injected[st]
                lm, _ = itm_loss(sd, ids, am, cond, cond_all, ids_all, mask_all, inj["neg_cond_idx"],
                                 inj["neg_text_idx"], cfg["itm_ratio"])
                l_itm.append(lm)
            out["loss_itc"] = sum(l_itc) / len(l_itc)
            out["loss_itm"] = sum(l_itm) / len(l_itm)
        elif t.startswith("cap"):
            ls = []
            for st in subtasks:
                cond = condition_feats(enc, st[1:])
                ls.append(cap_loss(sd, injected["cap"]["masked_ids"], am, injected["cap"]["labels"], cond))
            out["loss_cap"] = sum(ls) / len(ls)
        else:
            raise NotImplementedError(t)
    return out, enc


def token_masker(tokens, mask_prob, rng, mask_token=103, range_start=106, range_end=30522):
    """TokenMasker.perform_mask (data/model/general_module.py:64-97) with `rng` a random.Random: position 0 and
    pads (id 0) never masked, every row retried until >= 1 token is masked; 80% [MASK] / 10% random / 10% keep."""
    import numpy as np
    toks = np.array(tokens.cpu().numpy())
    ind = np.zeros(toks.shape, dtype=np.int64)
    for i in range(len(ind)):
        while all(ind[i] == 0):
            for j in range(1, len(ind[0])):
                if toks[i][j] != 0 and rng.random() < mask_prob:
                    ind[i][j] = 1
    labels = -np.ones(toks.shape, dtype=np.int64) * 100
    for i in range(toks.shape[0]):
        for j in range(toks.shape[1]):
            if ind[i][j] == 1:
                src = toks[i][j]
                p = rng.random()
                if p < 0.8:
                    toks[i][j] = mask_token
                elif p < 0.9:
                    toks[i][j] = rng.choice(list(range(range_start, range_end)))
                labels[i][j] = src
    return torch.from_numpy(toks).long(), torch.from_numpy(labels).long()


def token_masker_uniform(tokens, mask_prob, u_mask, u_kind, u_tok, mask_token=103, range_start=106, range_end=30522):
    """TokenMasker.perform_mask (data/model/general_module.py:64-97) with the uniform numbers given as tensors instead of drawn from
    `random` one by one (the rule of include/mico_hip.h: mico_token_mask): u_mask [rounds, b, S] - round r is used for a row only when
    rounds 0..r-1 selected nothing in it (the reference's retry loop); u_kind / u_tok [b, S] decide 80 % [MASK] / 10 % random id / 10 % keep."""
    import numpy as np
    toks = np.array(tokens.cpu().numpy())
    um, uk, ut = (np.asarray(x.detach().cpu().float().numpy()) for x in (u_mask, u_kind, u_tok))
    labels = -np.ones(toks.shape, dtype=np.int64) * 100
    pm = np.float32(mask_prob)
    for i in range(toks.shape[0]):
        ind = np.zeros(toks.shape[1], dtype=bool)
        for r in range(um.shape[0]):
            for j in range(1, toks.shape[1]):
                if toks[i][j] != 0 and um[r][i][j] < pm:
                    ind[j] = True
            if ind.any():
                break
        if not ind.any():
            # rounds exhausted: the reference keeps drawing until a token is masked (general_module.py:71); the uniform-number form forces
            # the floor(u_tok[i][0] * n)-th of the row's n maskable positions instead (include/mico_hip.h: mico_token_mask)
            cand = [j for j in range(1, toks.shape[1]) if toks[i][j] != 0]
            if cand:
                ind[cand[min(int(np.float32(ut[i][0]) * np.float32(len(cand))), len(cand) - 1)]] = True
        for j in range(toks.shape[1]):
            if ind[j]:
                src = toks[i][j]
                if uk[i][j] < np.float32(0.8):
                    toks[i][j] = mask_token
                elif uk[i][j] < np.float32(0.9):
                    toks[i][j] = min(range_start + int(np.float32(ut[i][j]) * np.float32(range_end - range_start)), range_end - 1)
                labels[i][j] = src
    return torch.from_numpy(toks).long(), torch.from_numpy(labels).long()


# --------------------------------------------------------------------------------------------------------------
# caption decoding (inference_demo.py:161-174).  The search itself is third-party code the reference tree does not
# hold: transformers==4.31.0 GenerationMixin.generate -> beam_search + BeamSearchScorer (set_env.sh:12 pins the
# version); it cannot run under the transformers 5.x of this image, so this restatement follows the published
# 4.31 algorithm and is NOT pinned by a reference-generated golden ("parity unpinned").
# The model-side protocol is the reference's own: bert.py:1110-1143 appends a [MASK] token per step, grows the
# 3-D attention mask causally and takes the logits of the last row; nothing is cached between steps.
# --------------------------------------------------------------------------------------------------------------
def grow_mask(m):
    """bert.py:1110-1117."""
    b, n, _ = m.shape
    out = m.new_zeros(b, n + 1, n + 1)
    out[:, :n, :n] = m
    out[:, n, :n] = m[:, n - 1, :n]
    out[:, n, n] = 1
    return out


def decode_step_logits(sd, ids, mask, cond, mask_token_id=103):
    step_ids = torch.cat([ids, torch.full((ids.shape[0], 1), mask_token_id, dtype=torch.long)], dim=1)
    seq = bert_forward(sd, step_ids, grow_mask(mask), cond)
    return bert_lm_logits(sd, seq[:, -1:])[:, 0]


def generate_beam(sd, cond, max_new_tokens, num_beams, length_penalty, bos=101, eos=102, pad=0, mask_token_id=103):
    """cond [B, E, 768] -> token ids [B, <= 1 + max_new_tokens] (first column [CLS]); see the block comment above."""
    B = cond.shape[0]
    nb = num_beams
    ids = torch.full((B * nb, 1), bos, dtype=torch.long)
    mask = torch.ones(B * nb, 1, 1, dtype=torch.long)
    cond_x = cond.repeat_interleave(nb, dim=0)
    running = torch.zeros(B, nb)
    running[:, 1:] = -1e9
    running = running.reshape(-1)
    finished = [[] for _ in range(B)]        # per sample: list of (score, ids), at most nb kept
    worst = [1e9] * B
    closed = [False] * B
    max_length = 1 + max_new_tokens

    def push(b, hyp, logp):
        score = logp / (hyp.shape[-1] ** length_penalty)
        if len(finished[b]) < nb or score > worst[b]:
            finished[b].append((score, hyp))
            if len(finished[b]) > nb:
                ranked = sorted((sc, i) for i, (sc, _) in enumerate(finished[b]))
                del finished[b][ranked[0][1]]
                worst[b] = ranked[1][0]
            else:
                worst[b] = min(score, worst[b])

    while True:
        logp = torch.log_softmax(decode_step_logits(sd, ids, mask, cond_x, mask_token_id).float(), dim=-1) + running[:, None]
        V = logp.shape[-1]
        cand_s, cand_i = torch.topk(logp.view(B, nb * V), 2 * nb, dim=1)
        new_s, new_t, new_src = torch.zeros(B, nb), torch.zeros(B, nb, dtype=torch.long), torch.zeros(B, nb, dtype=torch.long)
        length_now = ids.shape[1] + 1
        for b in range(B):
            if closed[b]:
                new_t[b] = pad
                continue
            filled = 0
            for rank in range(2 * nb):
                t = int(cand_i[b, rank]) % V
                src = b * nb + int(cand_i[b, rank]) // V
                if t == eos:
                    if rank < nb:
                        push(b, ids[src].clone(), float(cand_s[b, rank]))
                    continue
                new_s[b, filled], new_t[b, filled], new_src[b, filled] = cand_s[b, rank], t, src
                filled += 1
                if filled == nb:
                    break
            if len(finished[b]) >= nb and worst[b] >= float(cand_s[b].max()) / length_now ** length_penalty:
                closed[b] = True
        running = new_s.reshape(-1)
        ids = torch.cat([ids[new_src.reshape(-1)], new_t.reshape(-1, 1)], dim=1)
        mask = grow_mask(mask)
        if all(closed) or ids.shape[1] >= max_length:
            break
    picks = []
    for b in range(B):
        if not closed[b]:
            for k in range(nb):
                push(b, ids[b * nb + k], float(running[b * nb + k]))
        picks.append(sorted(finished[b], key=lambda h: h[0])[-1][1])
    width = min(max(len(h) for h in picks) + 1, max_length)
    out = torch.full((B, width), pad, dtype=torch.long)
    for b, h in enumerate(picks):
        out[b, :len(h)] = h
        if len(h) < width:
            out[b, len(h)] = eos
    return out


def generate_sample(sd, cond, max_new_tokens, top_k, noise, bos=101, eos=102, pad=0, mask_token_id=103, step_logits=None):
    """Top-k sampling decode of the reference's captioner_mode (data/model/vast.py:526-536: do_sample=True, top_k=10) under the
    transformers==4.31 `sample` loop semantics (TopKLogitsWarper -> softmax -> one draw per row; finished rows emit pad; stop when all
    rows are finished or at max_length).  Third-party arithmetic absent from the reference tree ("parity unpinned"): the draw is
    restated as inverse-CDF over the kept candidates in descending order with the injected uniform numbers noise[row, step]."""
    B = cond.shape[0]
    ids = torch.full((B, 1), bos, dtype=torch.long)
    mask = torch.ones(B, 1, 1, dtype=torch.long)
    alive = torch.ones(B, dtype=torch.bool)
    for step in range(max_new_tokens):
        # step_logits(ids, mask) substitutes another model step (tests: the product's own logits, so that the SEARCH is compared
        # bit-exactly while logit parity is gated separately - with random-init weights the top-k order flips within 16-bit rounding)
        logits = (step_logits(ids, mask) if step_logits is not None else decode_step_logits(sd, ids, mask, cond, mask_token_id)).float()
        top_s, top_i = torch.topk(logits, top_k, dim=-1)
        cdf = torch.softmax(top_s, -1).double().cumsum(-1)
        pick = (cdf < (noise[:, step].double() * cdf[:, -1])[:, None]).sum(-1).clamp_max(top_k - 1)
        tok = top_i[torch.arange(B), pick]
        tok = torch.where(alive, tok, torch.full_like(tok, pad))
        alive = alive & (tok != eos)
        ids = torch.cat([ids, tok[:, None]], 1)
        mask = grow_mask(mask)
        if not bool(alive.any()):
            break
    return ids


def itm_sample(sim, rank, bs, u):
    """Hard-negative draw of data/model/vast.py:423-440: weights = softmax(sim, 1) + 1e-4 with the own-rank diagonal zeroed, one
    multinomial draw per row.  The reference draws with torch.multinomial (global generator); restated here as inverse-CDF with the
    injected uniform numbers u[row] (same distribution).  Returns (indices, margin): margin[row] = distance of u * total to the nearest
    CDF edge relative to total - rows with a tiny margin may legitimately differ by one under another summation order."""
    w = torch.softmax(sim.float(), dim=1) + 1e-4
    r = torch.arange(bs)
    w[r, rank * bs + r] = 0
    cdf = w.double().cumsum(1)
    tgt = u.double() * cdf[:, -1]
    idx = (cdf <= tgt[:, None]).sum(1)          # first column whose CDF exceeds the target: never a zero-weight (diagonal) column
    n = sim.shape[1]
    last = torch.where(rank * bs + r == n - 1, n - 2, n - 1)
    idx = torch.where(idx >= n, last, idx)
    margin = (cdf - tgt[:, None]).abs().min(1).values / cdf[:, -1]
    return idx, margin
