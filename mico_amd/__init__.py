"""mico_amd - MI355X (gfx950) native implementation of MiCo's omni-modal forward/backward hot path.

Python host code (module surface of the reference: MiCo.from_pretrained, forward_*_encoder, pool_*, ...) on
PyTorch-ROCm for device memory / streams / torch.distributed, calling hand-written HIP kernels through the C-ABI
library libmico_hip.so (include/mico_hip.h).  No CPU fallback: see mico_amd/_lib.py.
"""
__version__ = "0.1.0"
