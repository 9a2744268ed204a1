"""hidvae_amd -- MI355X-native (gfx950) implementation of the HiD-VAE tokenizer training step.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all arithmetic on the hot path
is hand-written HIP behind the C ABI in include/hidvae.h, bound with ctypes in `_C.py`."""
import importlib
import os
import sys


def _hw_queue_cap(env=None):
    """hardware queues the HIP runtime creates per device: GPU_MAX_HW_QUEUES, 4 unless the user raised it"""
    env = os.environ if env is None else env
    try:
        return max(1, int(env.get("GPU_MAX_HW_QUEUES", "4")))
    except ValueError:
        return 4


def _graph_queues(env=None):
    """-> (value for DEBUG_HIP_FORCE_GRAPH_QUEUES or None = leave the runtime alone, list of warnings).  Pure: tests call it.

    What the knob is: the number of hardware queues hipGraphLaunch spreads a graph's parallel branches over.  The tagged step forks into
    one branch per level (plus the decoder on the caller's stream and RCCL's stream under data parallelism).  Measured on MI355X /
    ROCm 7.2 (bench.py, B = 1024, same box): the runtime's default 4 replays the tagged step in 1.79 ms -- two heavy branches end up
    behind each other --, 3 in 1.51 ms; the untagged step (one branch) is unchanged.
    What is NOT safe: more graph queues than the runtime has hardware queues (GPU_MAX_HW_QUEUES, default 4).  Round 3's suite died
    with a segmentation fault inside hipGraphLaunch at 5, 6 and 12 forced queues and passed at 8 once GPU_MAX_HW_QUEUES=8 was set with
    it (gpurun_out/r3c_tests.log, r3f_tests_q6.log, r3f_tests_q12.log, r3g_tests_q5.log, r3d_tests_nodp.log): every crash had the forced count
    above the hardware-queue count and no run at or below it ever crashed (the runtime's source is not in this image, so the mechanism
    inside hipGraphLaunch is not established beyond that).  So the value is CLAMPED to the cap, whoever asked for it -- HIDVAE_GRAPH_QUEUES or
    a DEBUG_HIP_FORCE_GRAPH_QUEUES already in the environment -- and the clamp is reported.
    HIDVAE_GRAPH_QUEUES=0 leaves the runtime's setting alone (a DEBUG_HIP_FORCE_GRAPH_QUEUES above the cap is still clamped)."""
    env = os.environ if env is None else env
    notes, cap = [], _hw_queue_cap(env)

    def parse(name, default):
        raw = env.get(name, default)
        try:
            return int(raw)
        except (TypeError, ValueError):
            notes.append(f"{name}={raw!r} is not an integer; using {default}")
            return int(default)

    explicit = env.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")
    if explicit is not None:
        q = parse("DEBUG_HIP_FORCE_GRAPH_QUEUES", "3")
        who = "DEBUG_HIP_FORCE_GRAPH_QUEUES"
    else:
        q = parse("HIDVAE_GRAPH_QUEUES", "3")
        who = "HIDVAE_GRAPH_QUEUES"
        if q == 0:
            return None, notes
    if q < 1:
        notes.append(f"{who}={q} is out of range; using 1")
        q = 1
    if q > cap:
        notes.append(f"{who}={q} exceeds the runtime's {cap} hardware queues (GPU_MAX_HW_QUEUES): hipGraphLaunch segfaults there; "
                     f"clamped to {cap} (raise GPU_MAX_HW_QUEUES with it if you mean it)")
        q = cap
    return str(q), notes


GRAPH_QUEUES = None  # what this import left in DEBUG_HIP_FORCE_GRAPH_QUEUES (None: the runtime's own default); bench.py records it


def _runtime_defaults():
    """HIP-runtime settings the step's graph wants, applied before the runtime initialises: it reads them at the first HIP call of the
    process, so `import hidvae_amd` must come before the first torch.cuda call; a later import cannot take effect and says so."""
    import warnings
    global GRAPH_QUEUES
    q, notes = _graph_queues()
    for n in notes:
        warnings.warn("hidvae_amd: " + n, RuntimeWarning, stacklevel=3)
    if q is not None:
        changed = os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES") != q
        tc = sys.modules.get("torch")
        if changed and tc is not None and getattr(tc, "cuda", None) is not None and tc.cuda.is_initialized():
            warnings.warn("hidvae_amd was imported after the HIP runtime initialised: the graph-queue setting "
                          f"(DEBUG_HIP_FORCE_GRAPH_QUEUES={q}) cannot take effect in this process and the tagged step replays ~20 % slower; "
                          "import hidvae_amd before the first torch.cuda call", RuntimeWarning, stacklevel=3)
        os.environ["DEBUG_HIP_FORCE_GRAPH_QUEUES"] = q
    GRAPH_QUEUES = os.environ.get("DEBUG_HIP_FORCE_GRAPH_QUEUES")


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
