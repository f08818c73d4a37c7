"""Deterministic synthetic parameters (there are no checkpoints or network in this environment).

Every tensor is generated from (seed, parameter name) alone, so the build container (golden fixture generation with
the reference), the CPU oracle and the GPU box all regenerate bit-identical weights without shipping them.
"""
import zlib

import torch


def synth_tensor(name, shape, seed=0):
    g = torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if name == "contra_temp":
        return torch.tensor(0.07)
    if name.endswith("logit_scale"):
        return torch.tensor(2.6592600)
    if len(shape) == 1 and name.endswith(".weight"):   # every 1-D ".weight" in this model is a LayerNorm gain
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if len(shape) <= 1:
        return 0.02 * torch.randn(shape, generator=g)
    return 0.02 * torch.randn(shape, generator=g)


SKIP_SUFFIXES = ("position_ids", "freqs_cos", "freqs_sin", "relative_position_index", "attn_mask")


def synth_state_dict(shapes, seed=0, tie_lm_head=True):
    """shapes: {name: shape}.  Buffers that are pure functions of the architecture (position ids, RoPE tables) are
    skipped.  cls.predictions.decoder.weight is tied to the word embeddings and decoder.bias to predictions.bias,
    as transformers==4.31 does for the reference (model/bert.py:1038-1041)."""
    sd = {}
    for name, shape in shapes.items():
        if name.endswith(SKIP_SUFFIXES):
            continue
        sd[name] = synth_tensor(name, shape, seed)
    if tie_lm_head:
        w = "multimodal_encoder.bert.embeddings.word_embeddings.weight"
        d = "multimodal_encoder.cls.predictions.decoder.weight"
        if w in sd and d in sd:
            sd[d] = sd[w]
        b, db = "multimodal_encoder.cls.predictions.bias", "multimodal_encoder.cls.predictions.decoder.bias"
        if b in sd and db in sd:
            sd[db] = sd[b]
    return sd


def synth_inputs(cfg, seed=1234):
    """Synthetic batch of the shapes SURVEY.md section 8d prescribes.  cfg: dict(b, image=n|0, video=n|0, audio=n|0, depth=n|0,
    S, res).  Returns CPU tensors."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    b, res, S = cfg["b"], cfg.get("res", 224), cfg.get("S", 77)
    out = {}
    nv = cfg.get("vision", 0)
    if nv:
        out["vision_pixels"] = torch.randn((b, nv, 3, res, res), generator=g)
    na = cfg.get("audio", 0)
    if na:
        out["audio_spectrograms"] = torch.randn((b, na, res, res), generator=g)
    nd = cfg.get("depth", 0)
    if nd:
        out["depth_pixels"] = torch.randn((b, nd, 3, res, res), generator=g)
    if S:
        ids = torch.randint(1000, 30000, (b, S), generator=g)
        lens = torch.randint(max(4, S // 3), S + 1, (b,), generator=g)
        lens[0] = S
        ar = torch.arange(S)[None]
        mask = (ar < lens[:, None]).long()
        ids[:, 0] = 101
        ids[torch.arange(b), lens - 1] = 102
        ids = ids * mask
        out["input_ids"] = ids
        out["attention_mask"] = mask
    return out
