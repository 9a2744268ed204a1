#!/usr/bin/env python3
"""Fixtures for the plain RqVae path, produced by running the REFERENCE's modules/rqvae.py in this
container with the same in-process shim as make_golden.py (stub gin, the spelling alias, TORCHDYNAMO_DISABLE=1):
    TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden_rqvae.py
Inputs and weights are oracle.fill formulas; a fixture holds the case description + the reference's outputs."""
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
import modules.loss as _L  # noqa: E402  (reference)

_L.CategoricalReconstuctionLoss = _L.CategoricalReconstructionLoss
from modules.rqvae import RqVae  # noqa: E402  (reference)
from modules.quantize import QuantizeForwardMode  # noqa: E402  (reference)

from oracle import torch_oracle as O  # noqa: E402

MODES = {O.STE: QuantizeForwardMode.STE, O.ROTATION: QuantizeForwardMode.ROTATION_TRICK}


class _B:
    pass


def run(name, cfg, B, training):
    P = O.formula_params(cfg, seed=100, with_tags=False)
    x, _, _ = O.formula_batch(cfg, B, seed=7, tagged=False)
    m = RqVae(input_dim=cfg.input_dim, embed_dim=cfg.embed_dim, hidden_dims=list(cfg.hidden_dims), codebook_size=cfg.codebook_size,
              codebook_kmeans_init=False, codebook_normalize=cfg.codebook_normalize, codebook_sim_vq=False,
              codebook_mode=MODES[cfg.codebook_mode], n_layers=cfg.n_layers, commitment_weight=cfg.commitment_weight, n_cat_features=cfg.n_cat_features)
    sd = m.state_dict()
    assert set(sd) == set(P), (sorted(sd), sorted(P))
    m.load_state_dict({k: v.clone() for k, v in P.items()})
    m.train(training)
    b = _B()
    b.x = x
    if training:
        out = m(b, gumbel_t=0.2)
        out.loss.backward()
    else:
        with torch.no_grad():
            out = m(b, gumbel_t=0.2)
    with torch.no_grad():
        q = m.get_semantic_ids(x, 0.2)
        margins = []
        for i, layer in enumerate(m.layers):
            res = q.residuals[:, :, i]
            cb = layer.out_proj(layer.embedding.weight)
            d = (res ** 2).sum(1, keepdim=True) + (cb.T ** 2).sum(0, keepdim=True) - 2 * res @ cb.T
            top2 = torch.topk(d, 2, dim=1, largest=False).values
            margins.append(top2[:, 1] - top2[:, 0])
    fx = {"sem_ids": q.sem_ids.numpy().astype(np.int32), "margins": torch.stack(margins, 1).numpy(),
          "embeddings": q.embeddings.numpy(), "residuals": q.residuals.numpy(), "quantize_loss": q.quantize_loss.numpy(),
          "embs_norm": out.embs_norm.numpy()}
    for k in ("loss", "reconstruction_loss", "rqvae_loss", "p_unique_ids"):
        fx[k] = np.float64(getattr(out, k).item())
    if training:  # codebook gradients in full; of the big MLP weights 64 strided samples + the norm (as make_golden.py does)
        norms = {}
        for k, p in m.named_parameters():
            norms[k] = float(p.grad.double().norm())
            if k.startswith("layers.") or p.grad.numel() <= 4096:
                fx["grad/" + k] = p.grad.numpy().copy()
            else:
                f = p.grad.detach().reshape(-1)
                fx["gsample/" + k] = f[::max(1, f.numel() // 64)][:64].numpy().copy()
        fx["grad_norms"] = json.dumps(norms)
    fx["desc"] = json.dumps(dict(name=name, cfg=cfg.__dict__, B=B, training=training, torch=torch.__version__))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
    print(f"{name:28s} loss={fx['loss']:.6f} p_unique={fx['p_unique_ids']:.3f} min_margin={fx['margins'].min():.2e}")


if __name__ == "__main__":
    torch.manual_seed(0)
    base = dict(commitment_weight=0.4, codebook_normalize=True, codebook_mode=O.ROTATION)
    run("rqvae_rot_train_b96", O.Cfg(**base), 96, True)
    run("rqvae_ste_train_nonorm_b48", O.Cfg(**{**base, "codebook_mode": O.STE, "codebook_normalize": False}), 48, True)
    run("rqvae_rot_eval_b64", O.Cfg(**base), 64, False)
    run("rqvae_rot_train_d64_b96", O.Cfg(**{**base, "embed_dim": 64}), 96, True)  # configs/rqvae_ml32m.gin:11
    run("rqvae_rot_train_ncat18_b48", O.Cfg(**{**base, "n_cat_features": 18}), 48, True)  # the ctor default (rqvae.py:50)
