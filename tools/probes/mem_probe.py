#!/usr/bin/env python
"""Where does the HBM go at the peak of the omni step?  Runs bench.py's omni workload (one rank's share of configs[3]) for two steps,
then one more step with the caching allocator's history on, and prints the live allocations at the end of the forward (= the start of the
backward, where the step peaks) grouped by the mico_amd source line that allocated them.
    python tools/probes/mem_probe.py [--workload omni] [--batch 64] [--diet N]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="omni")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--diet", type=int, default=None)
    a = ap.parse_args()
    from mico_amd import runtime
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict, synth_inputs
    dev = torch.device("cuda:0")
    bench.set_precision("fp16")
    runtime.set_activation_diet(a.diet)
    torch.manual_seed(0)
    model = MiCo(default_cfg("evaclip01_giant"))
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0), strict=False)
    model.to(dev).train()
    w = bench.WORKLOADS[a.workload]
    batch = {k: v.to(dev) for k, v in synth_inputs(dict(b=a.batch, **w["shape"]), seed=1234).items()}
    gib = 2.0 ** 30
    for it in range(3):
        model.zero_grad(set_to_none=True)
        if it == 2:
            torch.cuda.memory._record_memory_history(max_entries=400000, stacks="python")
            torch.cuda.reset_peak_memory_stats()
        losses = model(dict(batch), w["task"], compute_loss=True)
        total = sum(losses.values())
        if it == 2:
            torch.cuda.synchronize()
            print(f"end of forward: allocated {torch.cuda.memory_allocated() / gib:.1f} GiB, reserved {torch.cuda.memory_reserved() / gib:.1f} GiB, "
                  f"peak so far {torch.cuda.max_memory_allocated() / gib:.1f} GiB; plan {runtime.last_tower_plan}")
            snap = torch.cuda.memory._snapshot()
            by = collections.Counter()
            cnt = collections.Counter()
            for seg in snap["segments"]:
                for blk in seg["blocks"]:
                    if blk["state"] != "active_allocated":
                        continue
                    where = "(before history / no frame)"
                    frames = blk.get("frames") or []
                    chain = [f for f in frames if "/mico_amd/" in f["filename"] or f["filename"].endswith("bench.py")]
                    if chain:
                        where = " <- ".join(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in chain[:3])
                    by[where] += blk["size"]
                    cnt[where] += 1
            tot = sum(by.values())
            print(f"live at end of forward: {tot / gib:.1f} GiB in {sum(cnt.values())} blocks")
            for k, v in by.most_common(40):
                print(f"  {v / gib:8.2f} GiB  {cnt[k]:6d} blocks  {k}")
        if it == 2:
            # the backward, node by node: allocated at entry, peak inside, allocated at exit of every big autograd function (in execution order)
            from mico_amd import functional as Fn
            trace, saved = [], {}
            for cls in (Fn.EvaTowerFn, Fn.CrossKVFn, Fn.BertFn, Fn.LMHeadLossFn, Fn._CondPack):
                orig = saved[cls] = cls.backward

                def wrapped(ctx, *g, _orig=orig, _name=cls.__name__):
                    torch.cuda.synchronize()
                    a0 = torch.cuda.memory_allocated()
                    torch.cuda.reset_peak_memory_stats()
                    out = _orig(ctx, *g)
                    torch.cuda.synchronize()
                    trace.append((_name, a0, torch.cuda.max_memory_allocated(), torch.cuda.memory_allocated(), torch.cuda.memory_reserved()))
                    return out
                cls.backward = staticmethod(wrapped)
            peak_fwd = torch.cuda.max_memory_allocated()
        total.backward()
        torch.cuda.synchronize()
        if it == 2:
            for cls, orig in saved.items():
                cls.backward = staticmethod(orig)
            print("backward, node by node (GiB): allocated at entry / peak inside / allocated at exit / reserved at exit")
            for name, a0, pk, a1, rs in trace:
                print(f"  {name:14s} {a0 / gib:7.1f} {pk / gib:7.1f} {a1 / gib:7.1f} {rs / gib:7.1f}")
            print(f"forward peak {peak_fwd / gib:.1f} GiB")
            print(f"step peak {max([peak_fwd] + [t[2] for t in trace]) / gib:.1f} GiB (allocated; forward and the traced backward nodes), {torch.cuda.max_memory_reserved() / gib:.1f} GiB reserved")
            torch.cuda.memory._record_memory_history(enabled=None)


if __name__ == "__main__":
    main()
