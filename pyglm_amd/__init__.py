"""pyglm_amd -- MI355X (gfx950) Gibbs hot path of slinderman/pyglm behind the reference's Python API.

Host side: this package (Python on PyTorch-ROCm for device memory, streams and torch.distributed).
Device side: pyglm_amd/lib/libpyglm_hip.so (hand-written HIP, C ABI in include/pyglm_hip.h).
There is no CPU fallback: every numerical entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
