"""the ctypes stub a reference maintainer would add is extracted from the document and executed as
written (only the library path is substituted) - the documented binding must stay in step with the C-ABI."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_stub_runs(cuda):
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(import ctypes, torch\n.*?)```", doc, re.S)
    assert m, "the ctypes stub is missing from INTEGRATION.md"
    code = m.group(1).replace('ctypes.CDLL("libmico_hip.so")', f'ctypes.CDLL("{os.path.join(ROOT, "mico_amd", "libmico_hip.so")}")')
    import mico_amd._lib  # noqa: F401  (loads torch's HIP runtime before the library, as any torch program would have)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    ns["_lib"].mico_last_error_string.restype = __import__("ctypes").c_char_p
    x = torch.randn(300, 1408, device=cuda)
    w, b = torch.randn(1408, device=cuda), torch.randn(1408, device=cuda)
    for dt in (torch.bfloat16, torch.float16):
        y = ns["layer_norm_16"](x, w, b, 1e-6, out_dtype=dt)
        ref = F.layer_norm(x, (1408,), w, b, 1e-6)
        assert y.dtype == dt and ((y.float() - ref).abs().max() / ref.abs().max()).item() < (1e-2 if dt == torch.bfloat16 else 2e-3)
