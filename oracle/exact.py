"""TEST INFRASTRUCTURE ONLY (oracle/): ctypes front-end of oracle/exact.c (fixed-order fp32 restatement
of encoder + L-level residual quantisation).  See the header of exact.c for the algorithm citations
and the parity status (pinned to tests/golden by tests/test_exact_oracle.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "exact.c")
_OUT = os.path.join(_HERE, "_build", "libhidvae_oracle.so")
_lib = None


def build(force=False):
    """gcc-compile exact.c into oracle/_build/ (git-ignored; travels to the GPU box with gpurun)."""
    if not force and os.path.exists(_OUT) and os.path.getmtime(_OUT) >= os.path.getmtime(_SRC):
        return _OUT
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2",
                           "-fPIC", "-shared", "-o", _OUT, _SRC, "-lm"])
    return _OUT


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_OUT):
            build()
        _lib = ctypes.CDLL(_OUT)
        _lib.orc_exp.restype = ctypes.c_float
        _lib.orc_exp.argtypes = [ctypes.c_float]
        _lib.orc_silu.restype = ctypes.c_float
        _lib.orc_silu.argtypes = [ctypes.c_float]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _ptr_array(arrs):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def linear(x, w, silu=False):
    x, xp = _f(x)
    w, wp = _f(w)
    y = np.empty((x.shape[0], w.shape[0]), dtype=np.float32)
    lib().orc_linear(xp, ctypes.c_int64(x.shape[0]), ctypes.c_int64(x.shape[1]), wp, ctypes.c_int64(w.shape[0]),
                     y.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(1 if silu else 0))
    return y


def mlp(x, weights):
    """Encoder body without the final l2norm (that lives in rq_forward's prologue)."""
    h = np.ascontiguousarray(x, dtype=np.float32)
    for j, w in enumerate(weights):
        h = linear(h, w, silu=(j != len(weights) - 1))
    return h


def codebook_prepare(E, normalize):
    E, Ep = _f(E)
    cb = np.empty_like(E)
    cc = np.empty((E.shape[0],), dtype=np.float32)
    lib().orc_codebook_prepare(Ep, ctypes.c_int64(E.shape[0]), ctypes.c_int(int(normalize)),
                               cb.ctypes.data_as(ctypes.c_void_p), cc.ctypes.data_as(ctypes.c_void_p))
    return cb, cc


def rq_forward(y, codebooks, normalize_input, normalize_l0, mode, training, beta, cosine=False):
    """y [B,D] (pre-normalisation encoder output); codebooks: list of L raw [K,D] embedding tables (D = 32: ORDER-G/Q/P; else ORDER-GEN).
    Returns dict(z, ids[B,L] int64, emb_cat[B,L*32], emb_sum[B,32], res_cat[B,L*32], loss[B])."""
    y, yp = _f(y)
    if y.shape[1] != 32 or cosine:  # the width-independent kernels have their own operation order
        return rq_forward_gen(y, codebooks, normalize_input, normalize_l0, mode, training, beta, cosine)
    B, L = y.shape[0], len(codebooks)
    K = codebooks[0].shape[0]
    prepared = [codebook_prepare(E, normalize_l0 and i == 0) for i, E in enumerate(codebooks)]
    cbs = [p[0] for p in prepared]
    ccs = [p[1] for p in prepared]
    z = np.empty((B, 32), np.float32)
    ids = np.empty((B, L), np.int64)
    emb_cat = np.empty((B, L * 32), np.float32)
    emb_sum = np.empty((B, 32), np.float32)
    res_cat = np.empty((B, L * 32), np.float32)
    loss = np.empty((B,), np.float32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib().orc_rq_forward(yp, ctypes.c_int64(B), ctypes.c_int(int(normalize_input)), ctypes.c_int(L), ctypes.c_int64(K),
                         _ptr_array(cbs), _ptr_array(ccs), ctypes.c_int(mode), ctypes.c_int(int(training)),
                         ctypes.c_float(beta), vp(z), vp(ids), vp(emb_cat), vp(emb_sum), vp(res_cat), vp(loss))
    return dict(z=z, ids=ids, emb_cat=emb_cat, emb_sum=emb_sum, res_cat=res_cat, loss=loss, cbs=cbs, ccs=ccs)


def rq_forward_gen(y, codebooks, normalize_input, normalize_l0, mode, training, beta, cosine=False):
    """rq_forward at any embedding width D <= 64 (a multiple of 4): ORDER-GEN, the operation order of csrc/rq_generic.hip.
    y [B,D]; codebooks: list of L raw [K,D] tables.  Returns the same dict as rq_forward."""
    y, yp = _f(y)
    B, D = y.shape
    L, K = len(codebooks), codebooks[0].shape[0]
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    cbs, ccs = [], []
    for i, E in enumerate(codebooks):
        E, Ep = _f(E)
        if D == 32:  # (the cosine ranking at width 32: the tables still come from the width-32 prepare launch, in its order)
            cb, cc = codebook_prepare(E, normalize_l0 and i == 0)
            cbs.append(cb)
            ccs.append(cc)
            continue
        cb, cc = np.empty_like(E), np.empty((K,), np.float32)
        lib().orc_codebook_prepare_gen(Ep, ctypes.c_int64(K), ctypes.c_int(D), ctypes.c_int(int(normalize_l0 and i == 0)), vp(cb), vp(cc))
        cbs.append(cb)
        ccs.append(cc)
    z = np.empty((B, D), np.float32)
    ids = np.empty((B, L), np.int64)
    emb_cat, res_cat = np.empty((B, L * D), np.float32), np.empty((B, L * D), np.float32)
    emb_sum = np.empty((B, D), np.float32)
    loss = np.empty((B,), np.float32)
    lib().orc_rq_forward_gen(yp, ctypes.c_int64(B), ctypes.c_int(D), ctypes.c_int(int(normalize_input)), ctypes.c_int(L), ctypes.c_int64(K),
                             _ptr_array(cbs), _ptr_array(ccs), ctypes.c_int(mode), ctypes.c_int(int(training)), ctypes.c_float(beta),
                             vp(z), vp(ids), vp(emb_cat), vp(emb_sum), vp(res_cat), vp(loss), ctypes.c_int(int(cosine)))
    return dict(z=z, ids=ids, emb_cat=emb_cat, emb_sum=emb_sum, res_cat=res_cat, loss=loss, cbs=cbs, ccs=ccs)
