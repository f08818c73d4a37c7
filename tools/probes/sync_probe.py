#!/usr/bin/env python
"""Which calls of the omni step synchronise the host with the stream?  torch.cuda.set_sync_debug_mode("warn") makes every synchronising torch call
(blocking host <-> device copies, .item(), nonzero() ...) emit a warning; this prints each distinct call site once, with the mico_amd frames above it.
A sync in the middle of a step drains the queue the host has built up and leaves the GPU idle until the next launches arrive.
    python tools/probes/sync_probe.py [--layers N] [--direct]"""
import argparse
import collections
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--direct", action="store_true")
    a = ap.parse_args()
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict, synth_inputs
    dev = torch.device("cuda:0")
    bench.set_precision("fp16")
    torch.manual_seed(0)
    model = MiCo(default_cfg("evaclip01_giant", vision_layers=a.layers))
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=False)
    model.to(dev).train()
    w = bench.WORKLOADS["omni"]
    batch = {k: v.to(dev) for k, v in synth_inputs(dict(b=a.batch, **w["shape"]), seed=1234).items()}

    def step():
        model.zero_grad(set_to_none=True)
        losses = model(dict(batch), w["task"], compute_loss=True, backward_scale=None if a.direct else 1.0)
        sum(losses.values()).backward()

    step()
    step()
    torch.cuda.synchronize()
    sites = collections.Counter()
    first = {}

    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" not in str(message):
            return
        st = [f for f in traceback.extract_stack()[:-1] if "/mico_amd/" in f.filename or f.filename.endswith("sync_probe.py")]
        key = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(st[-4:]))
        sites[key] += 1
        first.setdefault(key, str(message)[:100])

    old = warnings.showwarning
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    try:
        step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
        warnings.showwarning = old
    torch.cuda.synchronize()
    print(f"{sum(sites.values())} synchronising calls in one step, {len(sites)} distinct sites:")
    for k, c in sites.most_common():
        print(f"  {c:4d} x  {k}\n          {first[k]}")


if __name__ == "__main__":
    main()
