"""EVA-CLIP vision tower behind the reference's factory surface (model/evaclip/factory.py:211-360)."""
from .eva_vit_model import EVAVisionTransformer, CustomCLIP, create_model, MODEL_CONFIGS  # noqa: F401
