"""Host logic of the optimizer row (f4) against values produced by the reference's own code (tests/golden/optimizer.pt, written
by `python -m oracle.make_golden opt`): learning-rate schedules and the parameter grouping of build_optimizer."""
import types

import torch
import torch.nn as nn

from common import golden


def test_schedules():
    from mico_amd import sched
    fx = golden("optimizer.pt")
    for name, want in fx["sched"].items():
        got = [getattr(sched, name)(x, 0.1) for x in fx["sched_x"]]
        assert got == want, name          # pure float arithmetic: identical expressions -> identical doubles
    o = types.SimpleNamespace(scheduler="warmup_linear", num_train_steps=200, warmup_ratio=0.05)
    assert [sched.get_lr_sched(st, o) for st in range(0, 201, 10)] == fx["get_lr_sched"]


def test_build_optimizer_grouping():
    from mico_amd import optim
    from mico_amd.model import AttrDict
    fx = golden("optimizer.pt")

    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.vision_encoder = nn.Module()
            self.vision_encoder.visual = nn.Module()
            self.vision_encoder.visual.proj = nn.Linear(3, 2)
            self.vision_encoder.visual.LayerNorm = nn.LayerNorm(2)
            self.multimodal_encoder = nn.Module()
            self.multimodal_encoder.dense = nn.Linear(2, 2)
            self.multimodal_encoder.LayerNorm = nn.LayerNorm(2)
            self.fresh_head = nn.Linear(2, 2)
            self.contra_temp = nn.Parameter(torch.tensor(0.07))

    tiny = Tiny()
    args = AttrDict(model_cfg=AttrDict(vision_encoder_type="evaclip01_giant"),
                    run_cfg=AttrDict(new_params_name=["fresh_head"], weight_decay=0.01, learning_rate=1e-4, new_lr=5e-4, clip_lr=5e-7,
                                     betas=[0.9, 0.98], optim="adamw"))
    opt = optim.build_optimizer(tiny, args, None)
    ids = {id(p): n for n, p in tiny.named_parameters()}
    got = [dict(names=[ids[id(p)] for p in g["params"]], lr=g["lr"], weight_decay=g["weight_decay"], init_lr=g["init_lr"])
           for g in opt.param_groups]
    assert got == fx["groups"]
    assert opt.new_params_name == fx["group_attrs"]["new_params_name"]
    assert opt.clip_lr_visual_len == fx["group_attrs"]["clip_lr_visual_len"]
    assert opt.defaults["eps"] == 1e-6 and opt.defaults["correct_bias"] is True
