"""CPU: the C-ABI library loads and exports every symbol include/hidvae.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hidvae.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hidvae_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import hidvae_amd
    from hidvae_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hidvae.h but not exported"
    bound = set(_C.exported_symbols())
    assert bound == set(names), f"ctypes table and header differ: {bound ^ set(names)}"
    assert b"gfx950" in ctypes.cast(ctypes.CDLL(_C.LIB_PATH).hidvae_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_product_refuses_cpu_tensors():
    import torch
    from hidvae_amd import _C
    with pytest.raises(RuntimeError):
        _C.gemm(_C.GEMM_NT, torch.zeros(4, 4), torch.zeros(4, 4))
