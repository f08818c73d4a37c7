#!/usr/bin/env python
"""Which ATen ops (with input shapes and Python callers) are behind the at::native kernels left on the omni step: one step of bench.py's
default workload under torch.profiler, ATen ops ranked by device time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mico_amd import runtime  # noqa: E402
from mico_amd.model import MiCo, default_cfg  # noqa: E402
from mico_amd.weights import synth_state_dict, synth_inputs  # noqa: E402

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["omni"]
bench.set_precision("fp16")
torch.manual_seed(0)
model = MiCo(default_cfg("evaclip01_giant", vision_layers=None))
model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=False)
model.to(dev).train()
b = int(sys.argv[1]) if len(sys.argv) > 1 else wl.get("batch", 64)
batch = {k: v.to(dev) for k, v in synth_inputs(dict(b=b, **wl["shape"]), seed=0).items()}


def step():
    model.zero_grad(set_to_none=True)
    losses = model(dict(batch), wl["task"], compute_loss=True)
    sum(losses.values()).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = [r for r in rows if r.key.startswith("aten::") and r.self_device_time_total > 0]
rows.sort(key=lambda r: -r.self_device_time_total)
for r in rows[:40]:
    stack = [s for s in r.stack if "mico_amd" in s or "bench.py" in s][:3]
    print(f"{r.key:28s} n={r.count:4d} dev_us={r.self_device_time_total:10.0f} shapes={str(r.input_shapes)[:110]}")
    for s in stack:
        print("      ", s[-120:])
