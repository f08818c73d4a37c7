#!/usr/bin/env python
"""Which 16-bit configuration holds north_star's 1e-3 gate?  Runs the ViT towers against the reference-generated goldens
(tests/golden/vit_*.pt) in: fp16 full split (x and W hi/lo, 3 k-segments), fp16 weights-only split (2 k-segments), plain fp16 (1), bf16.
Prints max|out - ref| / max|ref| of the final-LN tokens and the worst gradient-digest error per configuration.
    python tools/precision_probe.py [--full]      (--full adds the 40-block g/14 golden, forward only)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import golden, rel_err, build_model, grad_digest_check  # noqa: E402
from mico_amd import runtime  # noqa: E402

CONFIGS = [("fp16 split=full", torch.float16, True, "full"), ("fp16 split=weights", torch.float16, True, "weights"),
           ("fp16 plain", torch.float16, False, "full"), ("bf16", torch.bfloat16, False, "full")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--tail", default="", help="experiment: comma list of n:mode - split only the last n (first -n) blocks of the full g/14 tower")
    a = ap.parse_args()
    cuda = torch.device("cuda:0")
    for vtype, tag in (("evaclip02_base", "b16_d2"), ("evaclip01_giant", "g14_d2")):
        m, _ = build_model(vtype, 2, device=cuda)
        fx = golden(f"vit_{tag}.pt")
        g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
        x = torch.randn((2, 3, 224, 224), generator=g)
        w = torch.randn(fx["out"].shape, generator=g) / fx["out"].numel() ** 0.5
        for name, dt, split, mode in CONFIGS:
            runtime.CFG.split_fp16, runtime.CFG.split_mode = split, mode
            runtime.clear_weight_cache()
            m.zero_grad(set_to_none=True)
            with runtime.precision(dt):
                out = m.vision_encoder.visual(x.to(cuda), return_all_features=True)
                e = rel_err(out, fx["out"])
                (out * w.to(cuda)).sum().backward()
            named = dict(m.vision_encoder.visual.named_parameters())
            worst = max(grad_digest_check(d, named[n].grad, None) for n, d in fx["grads"].items())
            print(f"{tag:8s} {name:20s} fwd {e:.2e}   worst grad {worst:.2e}", flush=True)
        del m
    if a.full:
        from mico_amd.functional import l2_normalize
        fx = golden("vit_g14_full.pt")
        m, _ = build_model("evaclip01_giant", None, device=cuda)
        g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
        x = torch.randn((1, 1, 3, 224, 224), generator=g).to(cuda)
        for name, dt, split, mode in CONFIGS:
            runtime.CFG.split_fp16, runtime.CFG.split_mode = split, mode
            runtime.clear_weight_cache()
            with runtime.precision(dt), torch.no_grad():
                out = m.forward_vision_encoder(x)
                feat = l2_normalize(m.contra_head_v(m.pool_vision_for_contra(out)))
            e_rows = ((out[0, 0, [0, 1, 128, 256]].float().cpu() - fx["rows"]).abs().max() / fx["amax"]).item()
            print(f"g14_full {name:20s} token rows {e_rows:.2e}   feat_v {rel_err(feat, fx['feat_v']):.2e}", flush=True)
        from mico_amd import functional as Fn
        for item in [t for t in a.tail.split(",") if t]:
            n, mode = item.split(":")
            Fn._TAIL_EXP = (int(n), mode)
            runtime.clear_weight_cache()
            with runtime.precision(torch.float16), torch.no_grad():
                out = m.forward_vision_encoder(x)
                feat = l2_normalize(m.contra_head_v(m.pool_vision_for_contra(out)))
            Fn._TAIL_EXP = None
            e_rows = ((out[0, 0, [0, 1, 128, 256]].float().cpu() - fx["rows"]).abs().max() / fx["amax"]).item()
            print(f"g14_full tail {item:12s} token rows {e_rows:.2e}   feat_v {rel_err(feat, fx['feat_v']):.2e}", flush=True)
    runtime.CFG.split_fp16, runtime.CFG.split_mode = True, "full"


if __name__ == "__main__":
    main()
