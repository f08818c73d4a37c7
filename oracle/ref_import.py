"""TEST INFRASTRUCTURE ONLY - never imported by the product path (mico_amd/).

Imports the *reference* MiCo (read-only tree at /root/reference) on CPU inside the build container so that
golden fixtures can be generated from the reference's own code (oracle/make_golden.py) and so that the CPU
restatement in oracle/mico_oracle.py can be checked against it.  /root/reference does not exist on the GPU
box: nothing under tests/ -m gpu, bench.py or __graft_entry__.smoke() may import this module.

The reference cannot be imported as shipped: it needs torchvision / timm / easydict / ipdb
(absent here) and transformers==4.31 names that transformers 5.x moved.  The stand-ins below are import-time
name shims only - none of them executes on the forward/backward path that the goldens capture.
"""
import os
import sys
import types
import contextlib

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "model"))


def _install_shims():
    import torch.nn as nn
    import transformers  # must be imported before the torchvision stub is registered
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    import transformers.models.auto as au

    # transformers-4.31 names used by model/bert.py:29-35,57,60-61,881
    mu.apply_chunking_to_forward = pu.apply_chunking_to_forward
    mu.prune_linear_layer = pu.prune_linear_layer

    def _no_prune(*a, **k):
        raise NotImplementedError("head pruning is not on the MiCo hot path")

    mu.find_pruneable_heads_and_indices = _no_prune
    if not hasattr(au, "MODEL_FOR_VISION_2_SEQ_MAPPING"):
        au.MODEL_FOR_VISION_2_SEQ_MAPPING = {}
    for name, names in {
        "transformers.generation.beam_constraints": ["DisjunctiveConstraint", "PhrasalConstraint"],
        "transformers.generation.beam_search": ["BeamScorer", "BeamSearchScorer", "ConstrainedBeamSearchScorer"],
    }.items():
        if name not in sys.modules or not all(hasattr(sys.modules[name], n) for n in names):
            m = types.ModuleType(name)
            for n in names:
                setattr(m, n, type(n, (), {}))
            sys.modules[name] = m
    from transformers import PreTrainedModel

    if not hasattr(PreTrainedModel, "get_head_mask"):
        PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n

    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    class _Id:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    T = ["Normalize", "Compose", "RandomResizedCrop", "ToTensor", "Resize", "CenterCrop", "RandomHorizontalFlip"]
    if "torchvision" not in sys.modules:
        tv = mod("torchvision")
        tv.ops = mod("torchvision.ops")
        tv.ops.misc = mod("torchvision.ops.misc", FrozenBatchNorm2d=type("FrozenBatchNorm2d", (nn.Module,), {}))
        tv.transforms = mod(
            "torchvision.transforms",
            **{n: _Id for n in T},
            __all__=T + ["InterpolationMode"],
            InterpolationMode=types.SimpleNamespace(BICUBIC="bicubic", BILINEAR="bilinear"),
        )
        tv.transforms.functional = mod("torchvision.transforms.functional")
        tv.transforms.transforms = mod("torchvision.transforms.transforms", **{n: _Id for n in T}, __all__=T)
    if "ipdb" not in sys.modules:
        mod("ipdb", set_trace=lambda *a, **k: None)

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            for k, v in {**(d or {}), **kw}.items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            dict.__setattr__(self, k, v)
            dict.__setitem__(self, k, v)

        __setitem__ = __setattr__

    if "easydict" not in sys.modules:
        mod("easydict", EasyDict=EasyDict)
    if "timm" not in sys.modules:
        timm = mod("timm")
        timm.models = mod("timm.models")
        timm.models.layers = mod(
            "timm.models.layers",
            trunc_normal_=nn.init.trunc_normal_,
            DropPath=type("DropPath", (nn.Identity,), {}),
            to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
        )
    return EasyDict


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


_state = {}


def load():
    """Returns a namespace with the reference classes (MiCo, EasyDict, ...)."""
    if _state:
        return types.SimpleNamespace(**_state)
    if not available():
        raise RuntimeError("reference tree not present (this only works in the build container)")
    sys.dont_write_bytecode = True
    EasyDict = _install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with _cwd(REF_ROOT):
        from model.mico import MiCo  # noqa
        import model.bert as ref_bert  # noqa
        import model.evaclip.eva_vit_model as ref_eva  # noqa
    _state.update(MiCo=MiCo, EasyDict=EasyDict, ref_bert=ref_bert, ref_eva=ref_eva, cwd=lambda: _cwd(REF_ROOT))
    return types.SimpleNamespace(**_state)


def default_cfg(EasyDict, vision_encoder_type="evaclip01_giant", **over):
    cfg = dict(
        vision_encoder_type=vision_encoder_type, vision_resolution=224, checkpointing=False, contra_dim=512,
        max_vision_sample_num=8, max_audio_sample_num=4, max_depth_sample_num=1, frame_embedding_type="adaptive",
        pool_video=False, beam_size=3, itm_ratio=0.1, max_omni_caption_len=70, max_caption_len=40,
        max_subtitle_len=70,
    )
    cfg.update(over)
    return EasyDict(cfg)


def build_mico(vision_encoder_type="evaclip02_base", depth=None, **over):
    """Constructs the reference MiCo on CPU (random init); optionally truncates the ViT to `depth` blocks."""
    import torch
    ns = load()
    with ns.cwd():
        m = ns.MiCo.from_pretrained(default_cfg(ns.EasyDict, vision_encoder_type, **over), {})
    if depth is not None:
        m.vision_encoder.visual.blocks = torch.nn.ModuleList(list(m.vision_encoder.visual.blocks)[:depth])
    return m.eval()
