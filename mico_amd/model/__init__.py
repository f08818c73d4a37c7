from .mico import MiCo, MMGeneralModule, Contra_head, Match_head, AttrDict, TokenMasker  # noqa: F401
from .bert import BertForMaskedLM, build_tokenizer  # noqa: F401


def default_cfg(vision_encoder_type="evaclip01_giant", **over):
    """Model-cfg fields MiCo reads (defaults as data/caption_config/default_model_cfg.json)."""
    cfg = dict(vision_encoder_type=vision_encoder_type, vision_resolution=224, checkpointing=False, contra_dim=512,
               max_vision_sample_num=8, max_audio_sample_num=4, max_depth_sample_num=1, frame_embedding_type="adaptive",
               pool_video=False, beam_size=3, itm_ratio=0.1, max_omni_caption_len=70, max_caption_len=40, max_subtitle_len=70)
    cfg.update(over)
    return AttrDict(cfg)
