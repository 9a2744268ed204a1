"""Checkpoint files in the reference's layout (reference train_hidvae.py:1161-1171 writes, :621-628 and
modules/h_rqvae.py:382-471 read):

    {"iter", "model": state_dict, "model_config": HRqVae.config, "optimizer": torch.optim.AdamW.state_dict(),
     "accuracy", "rqvae_loss", "sem_id_repetition_rate"}

Interchange works in both directions.  The one pickled Python object in such a file is the enum
`modules.quantize.QuantizeForwardMode` inside model_config: a file written here names it by the REFERENCE's module path, so
the reference environment (which has no `hidvae_amd`) unpickles its own enum from it; a file written by the reference names the
same path, which this package resolves to its mirror while loading -- with or without install_dropin()."""
import contextlib
import sys
import types

import torch

from .modules import quantize as _quantize

_REF_MODULE = "modules.quantize"


@contextlib.contextmanager
def _reference_names():
    """While active, the reference's module names resolve to this package's mirrors (`modules.quantize.QuantizeForwardMode` IS this
    package's enum, for pickle's by-reference lookup in either direction; the other mirrored leaves are there because the
    reference's model_config also pickles the model object itself).  Whatever those names resolved to before is
    restored afterwards."""
    import importlib
    from . import _DROPIN, _PARENT_FALLBACK
    enum = _quantize.QuantizeForwardMode
    names = list(_PARENT_FALLBACK) + list(_DROPIN)
    saved = {k: sys.modules.get(k) for k in names}
    saved_mod = enum.__module__
    for parent in _PARENT_FALLBACK:  # bare stand-in packages: nothing but the mirrored leaves hangs off them
        pkg = types.ModuleType(parent)
        pkg.__path__ = []
        sys.modules[parent] = pkg
        if "." in parent:
            up, _, attr = parent.rpartition(".")
            setattr(sys.modules[up], attr, pkg)
    for theirs, ours in _DROPIN.items():
        mirror = importlib.import_module(f"hidvae_amd.{ours}")
        sys.modules[theirs] = mirror
        up, _, attr = theirs.rpartition(".")
        setattr(sys.modules[up], attr, mirror)
    enum.__module__ = _REF_MODULE
    try:
        yield
    finally:
        enum.__module__ = saved_mod
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def save_checkpoint(state: dict, path: str) -> None:
    with _reference_names():
        torch.save(state, path)


def load_checkpoint(path: str, map_location=None) -> dict:
    try:  # (a process that has the reference itself imported reads the reference's files natively)
        state = torch.load(path, map_location=map_location, weights_only=False)
    except (ModuleNotFoundError, AttributeError, ImportError):
        with _reference_names():
            state = torch.load(path, map_location=map_location, weights_only=False)
    cfg = state.get("model_config")
    if isinstance(cfg, dict):
        cfg.pop("self", None)  # the reference's config pickles the module itself; nothing reads it back
        cfg.pop("__class__", None)
        mode = cfg.get("codebook_mode")
        if mode is not None and not isinstance(mode, _quantize.QuantizeForwardMode):
            cfg["codebook_mode"] = _quantize.QuantizeForwardMode(getattr(mode, "value", mode))
    return state
