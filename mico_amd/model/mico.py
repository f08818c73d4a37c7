"""MiCo facade on the MI355X engine - drop-in surface of the reference's model/mico.py (class / method / parameter names,
argument meaning, error behaviour) plus `MiCo.forward(batch, task, compute_loss)` as specified by the VAST sibling
(data/model/vast.py:317-512).  All arithmetic runs in libmico_hip.so through
mico_amd.functional; there is no CPU path.
"""
import random

import torch
from torch import nn

from .. import functional as Fn
from .. import ops
from .bert import BertForMaskedLM, build_tokenizer
from .evaclip import create_model

VISION_TYPES = {   # mico.py:323-349
    "evaclip02_base": ("EVA02-CLIP-B-16", 768),
    "evaclip02_base_self": ("EVA02-CLIP-B-16", 768),
    "evaclip02_large": ("EVA02-CLIP-L-14", 1024),
    "evaclip01_giant": ("EVA01-CLIP-g-14", 1408),
    "evaclip02_bige": ("EVA02-CLIP-bigE-14-plus", 1792),   # the post-norm tower (mico.py:341-344)
}


class AttrDict(dict):
    """easydict-like config object (the reference passes an EasyDict as `opts`): attribute access to the keys, AttributeError for a
    missing / misspelled key exactly like EasyDict (so hasattr() means what it says); .get(key) for optional ones."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _cfg(opts):
    return opts if hasattr(opts, "vision_encoder_type") else AttrDict(opts)


class Contra_head(nn.Module):
    """mico.py:36-41."""

    def __init__(self, input_dim, contra_dim):
        super().__init__()
        self.linear = nn.Linear(input_dim, contra_dim, bias=False)

    def forward(self, cls_token):
        return Fn.linear_f32(cls_token, self.linear.weight, None)


class _Linear(nn.Linear):
    """nn.Linear parameter container evaluated with the exact-fp32 HIP GEMM (fused contra heads, mico.py:391-394)."""

    def forward(self, x):
        return Fn.linear_f32(x, self.weight, self.bias)


class Match_head(nn.Module):
    """mico.py:44-52: Linear -> GELU(erf) -> LayerNorm(1e-12) -> Linear(768 -> 2)."""

    def __init__(self, hidden_size):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, hidden_size)
        self.layernorm = nn.LayerNorm(hidden_size, eps=1e-12)
        self.linear2 = nn.Linear(hidden_size, 2)

    def forward(self, cls_token):
        cls_token = cls_token.float()   # the reference's `.half()` at vast.py:453 is an AMP artefact
        h = Fn.gelu_f32(Fn.linear_f32(cls_token, self.linear1.weight, self.linear1.bias))
        h = Fn.layer_norm_f32(h, self.layernorm.weight, self.layernorm.bias, 1e-12)
        return Fn.linear_f32(h, self.linear2.weight, self.linear2.bias)


class _HiddenTrans(nn.Sequential):
    """nn.Sequential(Linear(Dv, 768), LayerNorm(768, 1e-12)) container (mico.py:400-403); evaluated fused in cond_pack."""

    def __init__(self, din, dout):
        super().__init__(nn.Linear(din, dout), nn.LayerNorm(dout, eps=1e-12))

    def forward(self, x):
        zero = torch.zeros((1, self[0].out_features), dtype=torch.float32, device=x.device)
        y = Fn.cond_pack(x.reshape(-1, x.shape[-1]), self[0].weight, self[0].bias, self[1].weight, self[1].bias, zero, 1 << 30)
        return y.view(*x.shape[:-1], -1)


class TokenMasker:
    """data/model/general_module.py:52-97.  Device tokens (and no injected host RNG): one kernel (mico_token_mask) fed with torch.rand
    uniforms - no .cpu() copy, i.e. no stream sync in the middle of the step, and no b x S Python loop.  CPU tokens or an injected
    `random.Random`: the reference's host loops, draw for draw (what the golden fixtures and the oracle use).  `uniforms=(u_mask, u_kind,
    u_tok)` injects the device path's random numbers (parity against oracle.token_masker_uniform)."""

    # redraws of a row whose draw selected nothing (0.4^n per round for n maskable tokens).  When they run out the kernel forces one
    # maskable position (mico_token_mask), so every row with a maskable token leaves with >= 1 label, as general_module.py:71 guarantees.
    ROUNDS = 8

    def __init__(self, mask_token=103, range_start=106, range_end=30522, rng=None):
        self.mask_token, self.range = mask_token, (range_start, range_end)
        self.rng = rng or random
        self._host_rng_injected = rng is not None

    def __call__(self, tokens, mask_prob, uniforms=None):
        if tokens.is_cuda and (uniforms is not None or not self._host_rng_injected):
            if uniforms is None:
                b, S = tokens.shape
                u = torch.rand((self.ROUNDS + 2, b, S), device=tokens.device)
                uniforms = (u[:self.ROUNDS], u[self.ROUNDS], u[self.ROUNDS + 1])
            return ops.token_mask(tokens, mask_prob, uniforms[0].to(tokens.device), uniforms[1].to(tokens.device), uniforms[2].to(tokens.device),
                                  self.mask_token, self.range[0], self.range[1])
        toks = tokens.detach().cpu().clone().numpy()
        ind = [[0] * toks.shape[1] for _ in range(toks.shape[0])]
        for i in range(toks.shape[0]):
            while not any(ind[i]):
                for j in range(1, toks.shape[1]):
                    if toks[i][j] != 0 and self.rng.random() < mask_prob:
                        ind[i][j] = 1
        labels = torch.full(tokens.shape, -100, dtype=torch.long)
        for i in range(toks.shape[0]):
            for j in range(toks.shape[1]):
                if ind[i][j]:
                    src = int(toks[i][j])
                    p = self.rng.random()
                    if p < 0.8:
                        toks[i][j] = self.mask_token
                    elif p < 0.9:
                        toks[i][j] = self.rng.choice(range(*self.range))
                    labels[i][j] = src
        return torch.from_numpy(toks).long().to(tokens.device), labels.to(tokens.device)


class MMGeneralModule(nn.Module):
    """mico.py:61-371."""

    def construct_vision_encoder(self):
        t = self.config.vision_encoder_type
        if t.startswith("evaclip"):
            self.load_clip_model()
        elif t.startswith("swin"):
            self.load_swin_model()
        else:   # clip_* needs jit weights at absolute paths, the videoswin loader is undefined in the reference's model/
            raise NotImplementedError(f"vision_encoder_type {t!r} is not supported on the MI355X path")

    def load_swin_model(self):
        """mico.py:85-86 calls this, but the reference defines it only in the VAST sibling (data/model/general_module.py:528-578): the Swin-B /
        Swin-L 22k-224 towers of model/swin.py, built from the shape constants of their yaml files (pretrained weights are loaded by
        name through load_state_dict like every other checkpoint here)."""
        from .swin import SWIN_CONFIGS, SwinTransformer
        t = self.config.vision_encoder_type
        key = next((k for k in SWIN_CONFIGS if t.startswith(k)), None)
        if key is None:
            raise NotImplementedError(t)
        c = SWIN_CONFIGS[key]
        self.vision_encoder = SwinTransformer(img_size=self.config.vision_resolution, patch_size=4, in_chans=3, embed_dim=c["embed_dim"],
                                              depths=c["depths"], num_heads=c["num_heads"], window_size=7, drop_path_rate=c["drop_path_rate"])
        self.vision_dim = self.vision_encoder.num_features

    def load_clip_model(self):
        t = self.config.vision_encoder_type
        if t not in VISION_TYPES:
            raise NotImplementedError(t)
        name, self.vision_dim = VISION_TYPES[t]
        self.vision_encoder = create_model(name, force_custom_clip=True, image_size=self.config.vision_resolution,
                                           layers=self.config.get("vision_layers"))

    def construct_audio_encoder(self):
        self.audio_dim = self.vision_dim

    def construct_depth_encoder(self):
        self.depth_dim = self.vision_dim

    def construct_multimodal_encoder(self):
        self.multimodal_encoder = BertForMaskedLM()
        self.multimodal_dim = 768
        self.multimodal_encoder.tokenizer = build_tokenizer()

    # ---- encoders (mico.py:115-155) ----
    def forward_vision_encoder(self, vision_pixels):
        b, n, _, h, w = vision_pixels.shape
        if self.config.vision_encoder_type.startswith("swin"):      # mico.py:124-126
            out = self.vision_encoder(vision_pixels.reshape(b * n, 3, h, w))
            return out.reshape(b, -1, *out.shape[-2:])
        if not self.config.vision_encoder_type.startswith("evaclip"):
            raise NotImplementedError()
        out = self.vision_encoder.visual(vision_pixels.reshape(b * n, 3, h, w), return_all_features=True)
        return out.reshape(b, -1, *out.shape[-2:])

    def forward_audio_encoder(self, audio_spectrograms):
        # reference: unsqueeze(2).repeat(1,1,3,1,1) then the vision tower (mico.py:139-143); here the single channel meets
        # channel-summed patch weights - algebraically identical, no 3x copy.
        b, n, h, w = audio_spectrograms.shape
        if self.config.vision_encoder_type.startswith("swin"):      # the reference's own route: 3 identical channels (mico.py:139-140)
            return self.forward_vision_encoder(audio_spectrograms.unsqueeze(2).repeat(1, 1, 3, 1, 1))
        out = self.vision_encoder.visual.forward_groups([audio_spectrograms.reshape(b * n, 1, h, w)])
        return out.reshape(b, n, *out.shape[-2:])

    def forward_depth_encoder(self, depth_pixels):
        return self.forward_vision_encoder(depth_pixels)

    def forward_multimodal_encoder(self, input_ids, attention_mask, condition_feat=None, labels=None, position_ids=None,
                                   preprocess=True):
        return self.multimodal_encoder(input_ids=input_ids, attention_mask=attention_mask,
                                       encoder_hidden_states=condition_feat, labels=labels)

    # ---- pooling (mico.py:157-185) ----
    def pool_vision_for_contra(self, feature):
        if self.config.vision_encoder_type.startswith("swin"):      # no CLS token: token mean, then frame mean (mico.py:161-163)
            return Fn.mean_pool(feature)
        return Fn.cls_pool(feature)

    pool_audio_for_contra = pool_vision_for_contra
    pool_depth_for_contra = pool_vision_for_contra

    def pool_text_for_contra(self, feature):
        return feature[:, 0]

    # ---- condition packing (mico.py:187-248) ----
    def _pack(self, modality, feats):
        b, n, x, c = feats.shape
        if self.config.pool_video:
            feats = Fn.pool_video(feats)
            x = 2
        trans = getattr(self, f"hidden_trans_{modality}_multimodal")
        fe = getattr(self, f"{modality}_frame_embedding")
        if modality == "vision" and self.config.frame_embedding_type != "adaptive":   # mico.py:195-204: added for 'adaptive' only
            table = torch.zeros((n, self.multimodal_dim), dtype=torch.float32, device=feats.device)
        else:
            if n != fe.shape[1]:   # nearest interpolation of the frame slots (mico.py:196-200)
                idx = torch.floor(torch.arange(n, device=fe.device, dtype=torch.float32) * (fe.shape[1] / n)).long()
                fe = fe[:, idx]
            table = fe[0]
        table = table + getattr(self, f"{modality}_type_embeddings").view(1, -1)
        y = Fn.cond_pack(feats.reshape(b * n * x, c).contiguous(), trans[0].weight, trans[0].bias, trans[1].weight,
                         trans[1].bias, table, x)
        return y.view(b, n * x, self.multimodal_dim)

    def get_multimodal_forward_input_vision(self, vision_output):
        return self._pack("vision", vision_output)

    def get_multimodal_forward_input_audio(self, audio_output):
        return self._pack("audio", audio_output)

    def get_multimodal_forward_input_depth(self, depth_output):
        return self._pack("depth", depth_output)

    def get_multimodal_forward_input_subtitle(self, subtitle_output):
        b, s, c = subtitle_output.shape
        tr = self.hidden_trans_subtitle_multimodal
        y = Fn.cond_pack(subtitle_output.reshape(b * s, c).contiguous(), tr[0].weight, tr[0].bias, tr[1].weight, tr[1].bias,
                         self.subtitle_type_embeddings.view(1, -1), 1 << 30)
        return y.view(b, s, -1)

    # ---- checkpoint key remap + embedding interpolation (mico.py:250-321 == inference_demo.py:29-95) ----
    def modify_checkpoint(self, checkpoint):
        import torch.nn.functional as F
        new = {}
        for k, v in checkpoint.items():
            if "video" in k:
                new[k.replace("video", "vision")] = v
            elif "evaclip_model" in k:
                new[k.replace("evaclip_model", "vision_encoder")] = v
            elif "clip_model" in k:
                new[k.replace("clip_model", "vision_encoder")] = v
            else:
                new[k] = v.float()
        ck = new
        if self.config.frame_embedding_type == "adaptive":
            for key, n in (("vision_frame_embedding", self.config.max_vision_sample_num),
                           ("audio_frame_embedding", self.config.max_audio_sample_num)):
                if key in ck and ck[key].shape[1] != n:
                    ck[key] = F.interpolate(ck[key].permute(0, 2, 1), n, mode="nearest").permute(0, 2, 1)
        pk = "vision_encoder.visual.pos_embed"
        if self.config.vision_encoder_type.startswith("evaclip") and pk in ck:
            src = ck[pk][0]
            width = src.shape[-1]
            patch = ck["vision_encoder.visual.patch_embed.proj.weight"].shape[-1]
            grid = round((src.shape[0] - 1) ** 0.5)
            new_grid = self.config.vision_resolution // patch
            if new_grid != grid:
                oth = F.interpolate(src[1:].reshape(grid, grid, width).permute(2, 0, 1).unsqueeze(0), (new_grid, new_grid), mode="bilinear")
                oth = oth[0].permute(1, 2, 0).reshape(-1, width)
                ck[pk] = torch.cat((src[0:1], oth), dim=0).unsqueeze(0)
        return ck


class MiCo(MMGeneralModule):
    """mico.py:374-423 + forward() per vast.py."""

    def __init__(self, config):
        super().__init__()
        self.config = _cfg(config)
        self.construct_vision_encoder()
        self.construct_audio_encoder()
        self.construct_depth_encoder()
        self.construct_multimodal_encoder()
        c = self.config
        cd = c.contra_dim
        self.contra_head_t = Contra_head(self.multimodal_dim, cd)
        self.contra_head_s = Contra_head(self.multimodal_dim, cd)
        self.contra_head_v = Contra_head(self.vision_dim, cd)
        self.contra_head_a = Contra_head(self.audio_dim, cd)
        self.contra_head_d = Contra_head(self.depth_dim, cd)
        self.contra_head_va = _Linear(self.vision_dim + self.audio_dim, cd)
        self.contra_head_id = _Linear(self.vision_dim + self.depth_dim, cd)
        self.contra_head_vs = _Linear(self.vision_dim + self.multimodal_dim, cd)
        self.contra_head_vas = _Linear(self.vision_dim + self.audio_dim + self.multimodal_dim, cd)
        self.contra_temp = nn.Parameter(torch.tensor(0.07))
        self.itm_head = Match_head(self.multimodal_dim)
        md = self.multimodal_dim
        self.vision_frame_embedding = nn.Parameter(0.02 * torch.randn(1, c.max_vision_sample_num, md))
        self.audio_frame_embedding = nn.Parameter(0.02 * torch.randn(1, c.max_audio_sample_num, md))
        self.depth_frame_embedding = nn.Parameter(0.02 * torch.randn(1, c.max_depth_sample_num, md))
        self.hidden_trans_vision_multimodal = _HiddenTrans(self.vision_dim, md)
        self.hidden_trans_audio_multimodal = _HiddenTrans(self.audio_dim, md)
        self.hidden_trans_depth_multimodal = _HiddenTrans(self.depth_dim, md)
        self.hidden_trans_subtitle_multimodal = _HiddenTrans(md, md)
        self.vision_type_embeddings = nn.Parameter(0.02 * torch.randn(1, 1, md))
        self.audio_type_embeddings = nn.Parameter(0.02 * torch.randn(1, 1, md))
        self.depth_type_embeddings = nn.Parameter(0.02 * torch.randn(1, 1, md))
        self.subtitle_type_embeddings = nn.Parameter(0.02 * torch.randn(1, 1, md))
        self.beam_size = c.beam_size
        self.itm_ratio = c.itm_ratio
        self.max_omni_caption_len = c.max_omni_caption_len
        self.max_caption_len = c.max_caption_len
        self.max_subtitle_len = c.max_subtitle_len
        tok = self.multimodal_encoder.tokenizer
        self.text_masker = TokenMasker(mask_token=tok.mask_token_id, range_start=106, range_end=30522)

    @classmethod
    def from_pretrained(cls, opts, state_dict, *inputs, **kwargs):
        model = cls(opts, *inputs, **kwargs)
        missing_keys, unexpected_keys = model.load_state_dict(state_dict, strict=False)
        model.vision_encoder.text = None   # the reference deletes the CLIP text tower here (mico.py:419)
        if state_dict != {}:
            print(f"Unexpected keys {unexpected_keys}")
            print(f"missing_keys  {missing_keys}")
        return model

    # ------------------------------------------------------------------------------------------------------------------
    # forward(batch, task, compute_loss): task grammar "ret%tv%ta..._cap%tv..." (vast.py:317-348); sub-task letters
    # v = vision (image/video), a = audio, d = depth; fused conditions "va" (contra_head_va) and "vd" (contra_head_id).
    # ------------------------------------------------------------------------------------------------------------------
    from .mico_forward import forward, encode_batch, _feat_cond, _condition_feats   # noqa: E402,F401
