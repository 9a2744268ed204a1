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


def test_query_workspace_is_host_only_and_matches_the_documented_formulas():
    """hidvae_query_workspace needs no GPU: the sizes are what include/hidvae.h documents per entry point."""
    from hidvae_amd import _C
    q = _C.workspace_bytes
    assert q(_C.WS_GEMM, 1024, 512, 768, 1) == 0                       # direct path: no slabs
    assert q(_C.WS_GEMM, 65536, 512, 768, 4) == 4 * 65536 * 512 * 4   # LDS-tiled path with split_k slabs
    assert q(_C.WS_GEMM, 512, 768, 8192, 0) == 16 * 512 * 768 * 4     # deep-K weight gradient
    assert q(_C.WS_LINEAR_BWD, 128, 512, 768, 1) == 2 * 512 * 4          # small batch: the column-sum partials of the fallback
    sk = (4096 + 512 * 2 * 4096) * 4                                    # balanced kernel: arrival counters + two partial tiles per workgroup
    assert q(_C.WS_LINEAR_BWD, 1024, 512, 768, 1) == sk and q(_C.WS_LINEAR_BWD_ZEROED, 1024, 512, 768, 1) == 4096 * 4
    assert q(_C.WS_LINEAR_BWD_ZEROED, 1024, 128, 32, 1) == 0 and q(_C.WS_LINEAR_BWD_ZEROED, 32768, 512, 768, 0) == 0
    assert q(_C.WS_LINEAR_BWD, 32768, 512, 768, 0) == 16 * 512 * 768 * 4
    assert q(_C.WS_COLSUM, 1000, 48) == 16 * 48 * 4
    assert q(_C.WS_CODEBOOK_GRAD, 2048, 3, 256) == 0 and q(_C.WS_CODEBOOK_GRAD, 8192, 3, 256) == 3 * 256 * 4 * 32 * 4
    assert q(_C.WS_LAYERNORM_BWD_ALL, 1024, 230) == 2 * 256 * 230 * 4
    assert q(_C.WS_BATCHNORM_FWD, 1024, 512) == 3 * 16 * 512 * 4 and q(_C.WS_BATCHNORM_BWD, 1024, 512) == 2 * 32 * 512 * 4  # (64- / 32-row chunks)
    assert q(_C.WS_ID_CENSUS, 1024) == (4 * 1024 + 3) * 8
    assert q(_C.WS_TAG_LOSS, 128, 38) == (2 * 128 + 128 * 38) * 4
    with pytest.raises(RuntimeError):
        q(99, 1)
    with pytest.raises(RuntimeError):
        q(_C.WS_GEMM, 1, 2)  # too few dimensions
