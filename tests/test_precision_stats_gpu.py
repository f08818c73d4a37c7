"""The 1e-3 gate is a max-norm statistic of ONE input in the reference-generated golden; this test looks at the same statistic over
several random images (oracle = CPU fp32 restatement, pinned to that golden) for the precision configuration bench.py times."""
import os

import pytest
import torch

from common import build_model, rel_err
from mico_amd import runtime
from oracle import mico_oracle as O

pytestmark = pytest.mark.gpu


def _errors(m, sd, xs, cuda):
    from mico_amd.functional import l2_normalize
    out = []
    for x, (ref_tok, ref_feat) in xs:
        with torch.no_grad():
            tok = m.forward_vision_encoder(x.to(cuda))
            feat = l2_normalize(m.contra_head_v(m.pool_vision_for_contra(tok)))
        out.append((rel_err(tok, ref_tok), rel_err(feat, ref_feat)))
    return out


def test_bench_precision_over_inputs(cuda):
    torch.set_num_threads(32)
    n_img = int(os.environ.get("MICO_PRECISION_STATS_N", "4"))
    m, sd = build_model("evaclip01_giant", None, device=cuda)
    arch = O.ARCHS["evaclip01_giant"]
    xs = []
    for s in range(n_img):
        x = torch.randn((1, 1, 3, 224, 224), generator=torch.Generator().manual_seed(1000 + s))
        tok = O.forward_vision_encoder(sd, arch, x)
        feat = O.contra_feat(sd, "contra_head_v", O.pool_for_contra(tok))
        xs.append((x, (tok, feat)))
    import bench
    rows = {}
    for name in os.environ.get("MICO_PRECISION_STATS_CONFIGS", "fp16").split(","):
        bench.set_precision(name.split("@")[0])
        if "@" in name:   # experiment: fp16@6:full = the first 6 blocks in the full split mode
            n, mode = name.split("@")[1].split(":")
            runtime.CFG.head_split_blocks, runtime.CFG.head_split_mode = int(n), mode
        e = _errors(m, sd, xs, cuda)
        rows[name] = e
        print(f"{name:14s} tokens max {max(a for a, _ in e):.2e} mean {sum(a for a, _ in e) / len(e):.2e}   "
              f"feat_v max {max(b for _, b in e):.2e} mean {sum(b for _, b in e) / len(e):.2e}", flush=True)
    runtime.CFG.head_split_mode = "weights"
    bench.set_precision("fp16-split")
    runtime.set_compute_dtype(torch.bfloat16)
    e = rows["fp16"]
    # the timed configuration: every image under the gate on both tensors (measured over 8 images: tokens 7.9e-4 mean / 9.2e-4 max,
    # feat_v 7.8e-4 / 9.2e-4; plain fp16 everywhere: 8.8e-4 / 1.0e-3 and 9.2e-4 / 1.09e-3 - at the gate, like the reference's own fp16
    # autocast at 8.4e-4)
    assert sum(a for a, _ in e) / len(e) < 9e-4 and sum(b for _, b in e) / len(e) < 9e-4, e
    # every image under the gate itself (the bound used to sit at 1.05e-3).  Round 3 looked for a cheap configuration with
    # more margin (8 images: max tokens / feat_v): first 8 blocks weights-split 8.4e-4 / 8.9e-4, 16 blocks 8.2e-4 / 8.2e-4, LN-fed GEMMs only
    # (qkv + fc1) in 8 / 12 / 16 / 40 blocks 9.0e-4 / 1.14e-3 ... 8.1e-4 / 1.10e-3 (worse on feat_v), first 8 / 16 blocks in the 3-segment mode
    # 7.8e-4 / 8.7e-4 and 7.2e-4 / 6.9e-4 - the error follows the MFMA work spent, there is no cheap margin: fp16 operands round the
    # LayerNorm outputs, qkv, attention probabilities / outputs and the GELU output of every block, the weights are one source of seven.
    assert max(a for a, _ in e) < 1e-3 and max(b for _, b in e) < 1e-3, e
