"""hidvae_amd -- MI355X-native (gfx950) implementation of the HiD-VAE tokenizer training step.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the hot path
is hand-written HIP behind the C ABI in include/hidvae.h, bound with ctypes in `_C.py`."""
import importlib
import os
import sys


def _runtime_defaults():
    """HIP-runtime settings the step's graph wants, applied before the runtime initialises (it reads them at the first HIP call of the
    process, so `import hidvae_amd` must come before the first torch.cuda call; later is harmless but has no effect).

    DEBUG_HIP_FORCE_GRAPH_QUEUES: the number of hardware queues hipGraphLaunch spreads a graph's parallel branches over.  The tagged
    step forks into one branch per level (plus the decoder on the caller's stream and RCCL's stream under data parallelism).  Measured
    on MI355X / ROCm 7.2 (bench.py --tagged 1, B = 1024, same box): the runtime's default and an explicit 4 give 1.79 ms -- two of the
    heavy branches end up behind each other --, 3 queues 1.507-1.514 ms, 5 / 6 / 8 / 12 queues 1.52-1.56 ms; B = 2048: 2.81 -> 2.46 ms;
    the untagged step (one branch) is unchanged.  More than 4 is NOT safe: a process that instantiates many graphs (the GPU test
    suite) segfaults inside hipGraphLaunch with 5, 6 and 12 queues unless GPU_MAX_HW_QUEUES is raised too; 3 passes the whole suite.
    An explicit DEBUG_HIP_FORCE_GRAPH_QUEUES in the environment wins; HIDVAE_GRAPH_QUEUES=0 leaves the runtime's default alone."""
    q = os.environ.get("HIDVAE_GRAPH_QUEUES", "3")
    if q != "0":
        os.environ.setdefault("DEBUG_HIP_FORCE_GRAPH_QUEUES", q)


_runtime_defaults()

from . import _C  # noqa: F401,E402

# reference module name -> mirror module in this package.  Only LEAF modules are replaced: the reference's own parent packages
# (`modules`, `data`, `init`, `ops`, ...) stay what they are, so everything that is not mirrored (modules.utils, modules.model,
# modules.transformer.*, data.tags_processed, ...) still resolves to the reference's files.
_DROPIN = {
    "modules.h_rqvae": "modules.h_rqvae", "modules.rqvae": "modules.rqvae", "modules.encoder": "modules.encoder",
    "modules.quantize": "modules.quantize", "modules.loss": "modules.loss", "modules.normalize": "modules.normalize",
    "modules.tokenizer.h_semids": "modules.tokenizer.h_semids",
    "init.kmeans": "init.kmeans", "data.schemas": "data.schemas", "data.utils": "data.utils",
    "distributions.gumbel": "distributions.gumbel",
    "ops.triton.jagged": "ops_hip.jagged",  # (stage-2 imports padded_to_jagged_tensor)
}
# parent package -> the mirror package that stands in for it when the reference tree is NOT importable (then nothing
# un-mirrored could be resolved anyway)
_PARENT_FALLBACK = {"modules": "modules", "modules.tokenizer": "modules.tokenizer", "init": "init", "data": "data",
                    "distributions": "distributions", "ops": "ops_hip", "ops.triton": "ops_hip"}


def _dropin_parent(name):
    """the package object that `name` resolves to: the reference's own package when it is importable, else our mirror"""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except ModuleNotFoundError:
        pkg = importlib.import_module(f"hidvae_amd.{_PARENT_FALLBACK[name]}")
        sys.modules[name] = pkg
        if "." in name:
            up, _, attr = name.rpartition(".")
            setattr(_dropin_parent(up), attr, pkg)
        return pkg


def install_dropin(force=False):
    """Make the reference's module names (modules.h_rqvae, modules.quantize, data.schemas, ...) resolve to this package's
    mirrors, so the reference's own scripts and gin files pick up the HIP implementation without edits (INTEGRATION.md,
    section A).  Call it BEFORE importing the reference's scripts.  Only the mirrored leaf modules are replaced (sys.modules entry
    + attribute on the parent package); their parent packages remain the reference's, so `from modules.utils import parse_config`,
    `modules.model`, `modules.transformer.*` keep working."""
    for theirs in _DROPIN:
        if theirs in sys.modules and not force and not getattr(sys.modules[theirs], "__name__", "").startswith("hidvae_amd"):
            raise RuntimeError(f"{theirs} is already imported from elsewhere; call install_dropin() before importing the reference")
    for theirs, ours in _DROPIN.items():
        mirror = importlib.import_module(f"hidvae_amd.{ours}")
        up, _, attr = theirs.rpartition(".")
        parent = _dropin_parent(up)
        sys.modules[theirs] = mirror
        if parent is not mirror:
            setattr(parent, attr, mirror)
    return sorted(_DROPIN)


__all__ = ["_C", "install_dropin"]
