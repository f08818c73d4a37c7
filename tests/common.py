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
