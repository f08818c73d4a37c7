#!/usr/bin/env python
"""Which of the head blocks' forward GEMMs buy the timed precision its margin?  The full-depth g/14 golden (tests/golden/vit_g14_full.pt) under
plain fp16 with the first n blocks' weights hi/lo split in three ways: every forward GEMM ("weights", the timed default with n = 4), only the
LayerNorm-fed ones (qkv, fc1: "weights-ln", 74 % of a block's forward flops), only the ones that write the residual stream (attn.proj, fc2:
"weights-res", 26 %).  Prints max|out - ref| / max|ref| of four token rows and of feat_v.
    python tools/probes/head_split_probe.py [n:mode ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import golden, rel_err, build_model  # noqa: E402
from mico_amd import runtime  # noqa: E402
from mico_amd.functional import l2_normalize  # noqa: E402


def main():
    items = sys.argv[1:] or ["0:weights", "4:weights", "2:weights", "3:weights", "4:weights-ln", "8:weights-ln", "4:weights-res", "8:weights-res", "12:weights-res",
                             "16:weights-res", "40:weights-res"]
    cuda = torch.device("cuda:0")
    fx = golden("vit_g14_full.pt")
    m, _ = build_model("evaclip01_giant", None, device=cuda)
    g = torch.Generator().manual_seed(fx["meta"]["input_seed"])
    x = torch.randn((1, 1, 3, 224, 224), generator=g).to(cuda)
    runtime.CFG.split_fp16, runtime.CFG.split_mode = False, "full"
    for item in items:
        n, mode = item.split(":")
        runtime.CFG.head_split_blocks, runtime.CFG.head_split_mode = int(n), mode
        runtime.clear_weight_cache()
        with runtime.precision(torch.float16), torch.no_grad():
            out = m.forward_vision_encoder(x)
            feat = l2_normalize(m.contra_head_v(m.pool_vision_for_contra(out)))
        e_rows = ((out[0, 0, [0, 1, 128, 256]].float().cpu() - fx["rows"]).abs().max() / fx["amax"]).item()
        print(f"head split {item:16s} token rows {e_rows:.2e}   feat_v {rel_err(feat, fx['feat_v']):.2e}", flush=True)
    runtime.CFG.head_split_blocks, runtime.CFG.head_split_mode = 0, "weights"


if __name__ == "__main__":
    main()
