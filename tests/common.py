import contextlib
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def build_model(vtype, depth, device="cpu", seed=0, **over):
    """The product's host module with the deterministic synthetic weights (same generator as the golden fixtures)."""
    from mico_amd.model import MiCo, default_cfg
    from mico_amd.weights import synth_state_dict
    torch.manual_seed(0)
    m = MiCo(default_cfg(vtype, vision_layers=depth, **over))
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return m.to(device).eval(), sd


def grad_digest_check(digest, grad, tol):
    f = grad.detach().flatten().float().cpu()
    assert tuple(grad.shape) == tuple(digest["shape"])
    scale = digest["head"].abs().max().clamp_min(1e-20)
    e1 = ((f[:256] - digest["head"]).abs().max() / scale).item()
    e2 = abs(f.norm().item() - digest["norm"].item()) / max(digest["norm"].item(), 1e-20)
    return max(e1, e2)


PRECISION_CONFIGS = ("parity", "timed")


@contextlib.contextmanager
def precision_config(name):
    """The two fp16 configurations every golden test runs under:
    "parity" - forward GEMMs as x_hi W_hi + x_lo W_hi + x_hi W_lo (runtime default, 3 k-segments);
    "timed"  - EXACTLY what bench.py times (bench.set_precision("fp16")): plain fp16 MFMA operands everywhere, the forward GEMMs of the
               first bench.HEAD_SPLIT_BLOCKS tower blocks with hi/lo-split weights.  BERT, the heads and the losses are plain fp16."""
    import bench
    from mico_amd import runtime
    old = runtime.snapshot()
    try:
        if name == "parity":
            runtime.restore((torch.float16, True, "full", False, 0, "weights"))
        elif name == "timed":
            dt, split, mode, _ = bench.PRECISIONS["fp16"]
            runtime.restore((getattr(torch, dt), split, mode, False, bench.HEAD_SPLIT_BLOCKS, "weights"))
        else:
            raise KeyError(name)
        yield
    finally:
        runtime.restore(old)


class Errs:
    """Collects (name, error, tolerance) so that a failing run still prints every measured error before it asserts."""

    def __init__(self, tag):
        self.tag, self.rows = tag, []

    def add(self, name, err, tol):
        self.rows.append((name, float(err), tol))

    def check(self):
        for n, e, t in self.rows:
            print(f"[{self.tag}] {n:28s} {e:.3e}  (tol {t:g}){'   <-- OVER' if not e < t else ''}")
        bad = [(n, e, t) for n, e, t in self.rows if not e < t]
        assert not bad, (self.tag, bad)
