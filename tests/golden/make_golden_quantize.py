#!/usr/bin/env python3
"""Fixtures for ONE Quantize level used on its own with QuantizeDistance.COSINE (reference modules/quantize.py:115-119; HRqVae never
selects it, so the tokenizer fixtures cannot cover it), produced by running the REFERENCE's modules/quantize.py in this container
with the same in-process shim as make_golden.py (stub gin, TORCHDYNAMO_DISABLE=1):
    TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden_quantize.py
Inputs, tables and upstream gradients are oracle.fill formulas; a fixture holds the case description + the reference's outputs
(embeddings, ids, loss, top-2 margins of the ranking, and the gradients of  sum(emb * g_out) + sum(loss * g_loss))."""
import json
import os
import sys
import types

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

gin = types.ModuleType("gin")
gin.constants_from_enum = lambda c: c
gin.configurable = lambda f=None, **k: f if f is not None else (lambda g: g)
sys.modules["gin"] = gin
from modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode  # noqa: E402  (reference)

from oracle import fill  # noqa: E402

MODES = {2: QuantizeForwardMode.STE, 3: QuantizeForwardMode.ROTATION_TRICK}


def inputs(B, D, K, seed):
    """shared with tests/helpers.quantize_inputs"""
    x = fill.gauss((B, D), seed)
    E = fill.uniform((K, D), seed + 1, -1, 1) * fill.uniform((K, 1), seed + 2, 0.2, 1.5)  # rows of different lengths: cosine != L2 ranking
    g_out = fill.uniform((B, D), seed + 3, -1, 1)
    g_loss = fill.uniform((B,), seed + 4, 0.1, 1.0)
    return x, E.astype(np.float32), g_out, g_loss


def run(name, B, D, K, mode, training, normalize, beta=0.4, seed=70):
    x, E, g_out, g_loss = inputs(B, D, K, seed)
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, codebook_normalize=normalize, sim_vq=False, commitment_weight=beta,
                 forward_mode=MODES[mode], distance_mode=QuantizeDistance.COSINE)
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(E))
    q.train(training)
    xt = torch.from_numpy(x).requires_grad_(training)
    out = q(xt, temperature=0.2)
    fx = {"embeddings": out.embeddings.detach().numpy(), "ids": out.ids.numpy().astype(np.int32), "loss": out.loss.detach().numpy()}
    if training:
        ((out.embeddings * torch.from_numpy(g_out)).sum() + (out.loss * torch.from_numpy(g_loss)).sum()).backward()
        fx["grad_x"] = xt.grad.numpy()
        fx["grad_E"] = q.embedding.weight.grad.numpy()
    with torch.no_grad():
        cb = q.out_proj(q.embedding.weight)
        xd = xt.detach()
        d = -((xd / xd.norm(dim=1, keepdim=True)) @ cb.T / cb.T.norm(dim=0, keepdim=True))
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        fx["margins"] = (top2[:, 1] - top2[:, 0]).numpy()
        l2 = ((xd ** 2).sum(1, keepdim=True) + (cb.T ** 2).sum(0, keepdim=True) - 2 * xd @ cb.T).argmin(1)
    fx["desc"] = json.dumps(dict(name=name, B=B, D=D, K=K, mode=mode, training=training, normalize=normalize, beta=beta, seed=seed,
                                 torch=torch.__version__))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name:34s} loss.mean={fx['loss'].mean():.6f} min_margin={fx['margins'].min():.2e} "
          f"ids differing from the L2 ranking: {(l2 != out.ids).float().mean():.2f}")


def run_gumbel(name, B, D, K, normalize, beta=0.4, seed=90, temperature=0.2):
    """GUMBEL_SOFTMAX training with the cosine ranking (quantize.py:115-119,125-130); torch.rand inside distributions/gumbel.py:10 is
    routed to a formula so the draws can be replayed (fill.uniform(seed + 5))."""
    x, E, g_out, g_loss = inputs(B, D, K, seed)
    U = fill.uniform((B, K), seed + 5, 0.0, 1.0)
    q = Quantize(embed_dim=D, n_embed=K, do_kmeans_init=False, codebook_normalize=normalize, sim_vq=False, commitment_weight=beta,
                 forward_mode=QuantizeForwardMode.GUMBEL_SOFTMAX, distance_mode=QuantizeDistance.COSINE)
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(E))
    q.train(True)
    xt = torch.from_numpy(x).requires_grad_(True)
    saved = torch.rand
    torch.rand = lambda shape, **kw: torch.from_numpy(U).reshape(tuple(shape))
    try:
        out = q(xt, temperature=temperature)
    finally:
        torch.rand = saved
    ((out.embeddings * torch.from_numpy(g_out)).sum() + (out.loss * torch.from_numpy(g_loss)).sum()).backward()
    with torch.no_grad():
        cb = q.out_proj(q.embedding.weight)
        xd = xt.detach()
        d = -((xd / xd.norm(dim=1, keepdim=True)) @ cb.T / cb.T.norm(dim=0, keepdim=True))
        top2 = torch.topk(d, 2, dim=1, largest=False).values
    fx = {"embeddings": out.embeddings.detach().numpy(), "ids": out.ids.numpy().astype(np.int32), "loss": out.loss.detach().numpy(),
          "grad_x": xt.grad.numpy(), "grad_E": q.embedding.weight.grad.numpy(), "margins": (top2[:, 1] - top2[:, 0]).numpy(),
          "desc": json.dumps(dict(name=name, B=B, D=D, K=K, mode=1, training=True, normalize=normalize, beta=beta, seed=seed,
                                  temperature=temperature, torch=torch.__version__))}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name:34s} loss.mean={fx['loss'].mean():.6f} min_margin={fx['margins'].min():.2e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    if "--gumbel" in sys.argv:
        run_gumbel("qgumbel_cosine_train_b80", 80, 32, 128, False)
        run_gumbel("qgumbel_cosine_train_d64_norm_b48", 48, 64, 96, True, seed=91)
        sys.exit(0)
    run("quantize_cosine_rot_train_b200", 200, 32, 256, 3, True, False)
    run("quantize_cosine_ste_train_d64_b96", 96, 64, 100, 2, True, False)
    run("quantize_cosine_rot_eval_norm_b64", 64, 32, 256, 3, False, True)
