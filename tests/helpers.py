"""Shared fixture plumbing for the parity tests (CPU and GPU)."""
import glob
import json
import os

import numpy as np
import torch

from oracle import torch_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names(kind="case"):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        n = os.path.basename(p)[:-4]
        if kind == "rqvae" and n.startswith("rqvae"):
            out.append(n)
        elif kind == "quantize" and n.startswith("quantize"):
            out.append(n)
        elif kind == "qgumbel" and n.startswith("qgumbel"):
            out.append(n)
        elif kind == "case" and not n.startswith(("kmeans", "train", "rqvae", "tokenizer", "quantize", "qgumbel")):
            out.append(n)
        elif kind == "kmeans" and n.startswith("kmeans"):
            out.append(n)
        elif kind == "train" and n.startswith("train"):
            out.append(n)
    return out


def load(name):
    fx = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    desc = json.loads(str(fx["desc"]))
    return fx, desc


def quantize_inputs(desc):
    """inputs of the stand-alone Quantize fixtures (the same formulas as tests/golden/make_golden_quantize.py inputs())"""
    from oracle import fill
    B, D, K, seed = desc["B"], desc["D"], desc["K"], desc["seed"]
    x = fill.gauss((B, D), seed)
    E = (fill.uniform((K, D), seed + 1, -1, 1) * fill.uniform((K, 1), seed + 2, 0.2, 1.5)).astype(np.float32)
    return x, E, fill.uniform((B, D), seed + 3, -1, 1), fill.uniform((B,), seed + 4, 0.1, 1.0)


def cfg_of(desc):
    return O.Cfg(**desc["cfg"])


def inputs_of(desc):
    cfg = cfg_of(desc)
    P = O.formula_params(cfg, seed=desc.get("param_seed", 100), with_tags=True)
    x, te, ti = O.formula_batch(cfg, desc["B"], seed=desc.get("batch_seed", 7), tagged=desc["tagged"])
    if desc.get("invalid_level") is not None:
        ti[:, desc["invalid_level"]] = -1
    return cfg, P, x, te, ti


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def close(a, b, rtol, atol):
    """max|a-b| <= rtol * max|b| + atol  (atol covers gradients that are mathematically zero,
    e.g. a bias feeding a LayerNorm, where both sides hold only cancellation noise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return bool(np.abs(a - b).max() <= rtol * np.abs(b).max() + atol)


def sample(t, n=64):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].cpu().numpy().copy()


def zero_grad_keys(cfg):
    """Linear biases that feed a BatchNorm/LayerNorm directly: their gradient is mathematically 0, so
    what autograd leaves there is cancellation noise (~1e-10) that Adam then amplifies to ~lr per step
    (g/(sqrt(v)+eps) with |g| << eps).  No two implementations agree on those; compare them loosely."""
    ks = []
    for i in range(cfg.n_layers):
        if cfg.use_batch_norm:
            ks.append(f"tag_projectors.{i}.0.bias")
            for n in ("feature_extractor.0", "residual_block1.0", "residual_block2.0", "classifier.0"):
                ks.append(f"tag_predictors.{i}.{n}.bias")
        if cfg.codebook_normalize:
            ks.append(f"tag_projectors.{i}.4.bias")
    return set(ks)


NEAR_TIE = 1e-6  # a top-2 distance gap below this is inside fp32 summation-order noise: that decision may legitimately flip


def safe_rows(fx, thr=NEAR_TIE):
    """[B] bool: items none of whose level decisions is a near-tie in the reference's own arithmetic.  An item with a near-tie at
    level i is excluded from level i on (a flip there changes every later residual of that item); tests compare ids / per-item
    floats on the safe rows bit-exactly / to tolerance and REPORT what happened on the others instead of hiding it."""
    return (np.asarray(fx["margins"]) > thr).all(axis=1)


def report_flips(name, got_ids, fx, thr=NEAR_TIE):
    """-> number of near-tie items whose id tuple differs from the reference's (printed with their margins; pytest -s shows it)"""
    safe = safe_rows(fx, thr)
    ref = np.asarray(fx["sem_ids"]).astype(np.int64)
    bad = np.nonzero(~safe & (np.asarray(got_ids) != ref).any(axis=1))[0]
    for b in bad:
        print(f"[near-tie flip] {name}: item {b} margins {np.asarray(fx['margins'])[b]} got {np.asarray(got_ids)[b]} reference {ref[b]}")
    if (~safe).any():
        print(f"[near-tie census] {name}: {int((~safe).sum())} item(s) with a top-2 gap <= {thr:g}; {len(bad)} flipped")
    return len(bad)


def grad_rtol(base, B):
    """Relative tolerance for a GRADIENT compared across implementations at batch size B.  A parameter gradient is a sum of B
    per-item terms of mixed sign accumulated in fp32 in an implementation-specific order; its rounding noise grows like sqrt(B).
    `base` (2e-5 oracle-vs-reference, 3e-5 HIP-vs-reference) is the bar the fixtures up to B = 256 meet; larger batches get
    base * sqrt(B / 256).  Losses and ids are NOT relaxed (1e-5 / bit-exact at every size)."""
    return base * max(1.0, (B / 256.0) ** 0.5)
