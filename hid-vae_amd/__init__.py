"""hidvae_amd -- MI355X-native (gfx950) implementation of the HiD-VAE tokenizer training step.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the hot path
is hand-written HIP behind the C ABI in include/hidvae.h, bound with ctypes in `_C.py`."""
import importlib
import sys

from . import _C  # noqa: F401

_DROPIN = {
    "modules": "modules", "modules.h_rqvae": "modules.h_rqvae", "modules.rqvae": "modules.rqvae", "modules.encoder": "modules.encoder",
    "modules.quantize": "modules.quantize", "modules.loss": "modules.loss", "modules.normalize": "modules.normalize",
    "modules.tokenizer": "modules.tokenizer", "modules.tokenizer.h_semids": "modules.tokenizer.h_semids",
    "init": "init", "init.kmeans": "init.kmeans", "data.schemas": "data.schemas", "data.utils": "data.utils",
    "ops": "ops_hip", "ops.triton": "ops_hip", "ops.triton.jagged": "ops_hip.jagged",  # (stage-2 imports padded_to_jagged_tensor)
}


def install_dropin(force=False):
    """Alias the reference's module names (modules.h_rqvae, data.schemas, ...) to this package's mirrors in sys.modules, so
    the reference's own scripts and gin files pick up the HIP implementation without edits.
    `data` itself is left alone (its ingestion modules are not mirrored); only data.schemas / data.utils are aliased."""
    for theirs, ours in _DROPIN.items():
        if theirs in sys.modules and not force and not getattr(sys.modules[theirs], "__name__", "").startswith("hidvae_amd"):
            raise RuntimeError(f"{theirs} is already imported from elsewhere; call install_dropin() before importing the reference")
        sys.modules[theirs] = importlib.import_module(f"hidvae_amd.{ours}")
    return sorted(_DROPIN)


__all__ = ["_C", "install_dropin"]
