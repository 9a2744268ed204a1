"""hidvae_amd -- MI355X-native (gfx950) implementation of the HiD-VAE tokenizer training step.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the hot path
is hand-written HIP behind the C ABI in include/hidvae.h, bound with ctypes in `_C.py`."""
from . import _C  # noqa: F401

__all__ = ["_C"]
