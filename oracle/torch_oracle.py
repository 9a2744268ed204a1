"""TEST INFRASTRUCTURE ONLY (oracle/): fp32 torch-CPU restatement of the HiD-VAE tokenizer
training step.  It is the *checker* for the HIP path and the `cpu_baseline` ("port") leg of
bench.py -- never the thing shipped: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this package.  The product (hid-vae_amd/) never does.

Parity status: PINNED by outputs of the reference itself, captured in this container by
tests/golden/make_golden.py (reference imported eagerly through the gin/typo shim of
SURVEY.md section 8c) and committed as tests/golden/*.npz.  tests/test_oracle_golden.py
replays every fixture through this file.

Style: pure functions over a flat {state_dict_key: tensor} mapping, so the same weights feed the
reference (load_state_dict), this oracle and the HIP path.  Each function cites the reference
lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

GUMBEL, STE, ROTATION = 1, 2, 3  # modules/quantize.py:17-20


@dataclass
class Cfg:
    """Constructor surface of HRqVae (modules/h_rqvae.py:231-256) plus the loss knobs the
    trainer pokes onto model.tag_prediction_loss (train_hidvae.py:526-530)."""
    input_dim: int = 768
    embed_dim: int = 32
    hidden_dims: List[int] = field(default_factory=lambda: [512, 256, 128])
    codebook_size: int = 256
    n_layers: int = 3
    codebook_normalize: bool = True
    codebook_sim_vq: bool = False
    codebook_mode: int = ROTATION
    commitment_weight: float = 0.25
    tag_alignment_weight: float = 0.5
    tag_prediction_weight: float = 0.5
    tag_class_counts: Optional[List[int]] = None
    tag_embed_dim: int = 768
    use_focal_loss: bool = False
    focal_loss_params: Optional[Dict] = None
    dropout_rate: float = 0.2
    use_batch_norm: bool = True
    alignment_temperature: float = 0.1
    sem_id_uniqueness_weight: float = 0.5
    sem_id_uniqueness_margin: float = 0.5
    use_label_smoothing: bool = True
    label_smoothing_alpha: float = 0.1
    use_mixup: bool = True
    mixup_alpha: float = 0.2
    n_cat_features: int = 0  # the ctor default is 18 (h_rqvae.py:243); every shipped config binds 0

    def classes(self):
        c = self.tag_class_counts if self.tag_class_counts is not None else [10, 100, 1000]
        return list(c[: self.n_layers])  # h_rqvae.py:279-283


# ----------------------------------------------------------------------------------------
# randomness providers: the reference draws dropout masks, the mixup permutation / lambda and
# Gumbel noise from torch's global CPU stream; a device stream can never match that, so every
# implementation (reference-under-patch, oracle, HIP path) takes them from a provider.
# ----------------------------------------------------------------------------------------
class FormulaRand:
    """Masks / permutations from oracle.fill hashes, numbered in call order."""

    def __init__(self, base_seed=5000, lam=0.3):
        from . import fill
        self._fill = fill
        self.base = base_seed
        self.lam = lam
        self.n_drop = 0
        self.n_mix = 0
        self.n_gum = 0

    def dropout_keep(self, shape, p):
        m = self._fill.keep_mask(tuple(shape), self.base + self.n_drop, p)
        self.n_drop += 1
        return torch.from_numpy(m)

    def mixup(self, n):
        p = torch.from_numpy(self._fill.perm(n, self.base + 700 + self.n_mix))
        self.n_mix += 1
        return p, torch.tensor(self.lam, dtype=torch.float32)

    def gumbel_u(self, shape):
        u = torch.from_numpy(self._fill.uniform(tuple(shape), self.base + 900 + self.n_gum, 0.0, 1.0))
        self.n_gum += 1
        return u


class TorchRand:
    """Same draws the reference makes from torch's global generator (used for CPU timing)."""

    def __init__(self, mixup_alpha=0.2):
        self.alpha = mixup_alpha

    def dropout_keep(self, shape, p):
        return torch.empty(tuple(shape)).bernoulli_(1 - p)

    def mixup(self, n):
        lam = torch.distributions.Beta(torch.tensor(self.alpha), torch.tensor(self.alpha)).sample()
        return torch.randperm(n), lam

    def gumbel_u(self, shape):
        return torch.rand(tuple(shape))


def _dropout(x, p, training, rand):
    # nn.Dropout in train mode == x * (bernoulli(1-p) / (1-p))   (ATen native dropout)
    if not training or p == 0.0:
        return x
    keep = rand.dropout_keep(x.shape, p)
    return x * (keep / (1.0 - p))


# ----------------------------------------------------------------------------------------
# a2/a3  MLP  (modules/encoder.py:23-36)
# ----------------------------------------------------------------------------------------
def mlp(x, weights, normalize):
    h = x
    for j, w in enumerate(weights):
        h = F.linear(h, w)
        if j != len(weights) - 1:
            h = F.silu(h)
    if normalize:
        h = F.normalize(h, p=2, dim=-1, eps=1e-12)  # modules/normalize.py:7-8
    return h


def enc_weights(P, cfg):
    return [P[f"encoder.mlp.{2 * j}.weight"] for j in range(len(cfg.hidden_dims) + 1)]


def dec_weights(P, cfg):
    return [P[f"decoder.mlp.{2 * j}.weight"] for j in range(len(cfg.hidden_dims) + 1)]


# ----------------------------------------------------------------------------------------
# a4/a5/a6  one quantisation level  (modules/quantize.py:100-153, :34-45)
# ----------------------------------------------------------------------------------------
def effective_codebook(P, cfg, i):
    cb = P[f"layers.{i}.embedding.weight"]
    if cfg.codebook_sim_vq:
        cb = F.linear(cb, P[f"layers.{i}.out_proj.0.weight"])  # quantize.py:70-73
    if i == 0 and cfg.codebook_normalize:  # h_rqvae.py:295
        cb = F.normalize(cb, p=2, dim=-1, eps=1e-12)
    return cb


def distances(x, cb):
    # quantize.py:109-113: |x|^2 + |c|^2 - (2x) c^T, in that association order
    return (x ** 2).sum(dim=1, keepdim=True) + (cb.T ** 2).sum(dim=0, keepdim=True) - 2 * x @ cb.T


def rotation_out(x, e):
    # quantize.py:34-45 with u,q,w detached; x is differentiated through
    u = (x / (x.norm(dim=-1, keepdim=True) + 1e-8)).detach()
    q = (e / (e.norm(dim=-1, keepdim=True) + 1e-8)).detach()
    w = F.normalize(u + q, p=2, dim=1, eps=1e-6).detach()
    xw = (x * w).sum(-1, keepdim=True)
    xu = (x * u).sum(-1, keepdim=True)
    return x - 2 * xw * w + 2 * xu * q


def cosine_distances(x, cb):
    # quantize.py:115-119: -(x / |x|) c^T / |c|, no epsilons
    return -((x / x.norm(dim=1, keepdim=True)) @ cb.T / cb.T.norm(dim=0, keepdim=True))


def quantize_level(x, cb, mode, beta, training, temperature, rand, cosine=False):
    dist = cosine_distances(x, cb) if cosine else distances(x, cb)
    ids = dist.detach().min(dim=1).indices  # first minimum on CPU
    if training:
        if mode == GUMBEL:
            u = rand.gumbel_u(dist.shape)  # distributions/gumbel.py:8-18
            g = -torch.log(-torch.log(u + 1e-20) + 1e-20)
            wts = F.softmax((-dist + g) / temperature, dim=-1)
            emb = wts @ cb
            out = emb
        elif mode == STE:
            emb = cb[ids]
            out = x + (emb - x).detach()
        elif mode == ROTATION:
            emb = cb[ids]
            out = rotation_out(x, emb)
        else:
            raise ValueError("Unsupported Quantize forward mode.")
        val = emb
    else:
        out = cb[ids]
        val = out
    # modules/loss.py:41-44
    loss = ((x.detach() - val) ** 2).sum(-1) + beta * ((x - val.detach()) ** 2).sum(-1)
    return out, ids, loss, dist


# ----------------------------------------------------------------------------------------
# a9  tag projector  (h_rqvae.py:322-331)   a10  tag predictor (h_rqvae.py:112-227)
# ----------------------------------------------------------------------------------------
def tag_projector(P, cfg, i, t, training, rand, bn_buffers=None):
    pre = f"tag_projectors.{i}."
    h = F.linear(t, P[pre + "0.weight"], P[pre + "0.bias"])
    if cfg.use_batch_norm:
        rm = rv = None
        if bn_buffers is not None:
            rm, rv = bn_buffers[pre + "1.running_mean"], bn_buffers[pre + "1.running_var"]
        elif not training:
            rm, rv = P[pre + "1.running_mean"], P[pre + "1.running_var"]
        h = F.batch_norm(h, rm, rv, P[pre + "1.weight"], P[pre + "1.bias"], training=training,
                         momentum=0.1, eps=1e-5)
    h = F.relu(h)
    h = _dropout(h, cfg.dropout_rate, training, rand)
    h = F.linear(h, P[pre + "4.weight"], P[pre + "4.bias"])
    if cfg.codebook_normalize:
        h = F.layer_norm(h, h.shape[-1:], P[pre + "5.weight"], P[pre + "5.bias"], 1e-5)
    return h


def predictor_dims(cfg, i):
    e = cfg.embed_dim * (i + 1)
    hidden = cfg.hidden_dims[0] // 2 * (i + 1)  # h_rqvae.py:314
    mid = int(hidden * 0.9)  # :151
    p = min(0.55, cfg.dropout_rate + i * 0.075)  # :129
    return e, hidden, mid, p


def tag_predictor(P, cfg, i, x, training, rand):
    pre = f"tag_predictors.{i}."
    e, hidden, mid, p = predictor_dims(cfg, i)
    ln = cfg.use_batch_norm  # LayerNorms are gated on use_batch_norm (h_rqvae.py:145)

    def lin(h, name):
        return F.linear(h, P[pre + name + ".weight"], P[pre + name + ".bias"])

    def lnorm(h, name):
        if not ln:
            return h
        return F.layer_norm(h, h.shape[-1:], P[pre + name + ".weight"], P[pre + name + ".bias"], 1e-5)

    a = torch.sigmoid(lin(F.gelu(lin(F.relu(lin(x, "attention.0")), "attention.2")), "attention.4"))
    h = x * a
    if i > 0:  # :194,211
        h = F.normalize(h, p=2, dim=-1)
    f = _dropout(F.relu(lnorm(lin(h, "feature_extractor.0"), "feature_extractor.1")), p, training, rand)
    for rb in ("residual_block1", "residual_block2"):
        r = _dropout(F.relu(lnorm(lin(f, rb + ".0"), rb + ".1")), p, training, rand)
        r = _dropout(F.relu(lin(r, rb + ".4")), p, training, rand)
        r = lnorm(r, rb + ".7")
        f = f + r
    c = _dropout(F.relu(lnorm(lin(f, "classifier.0"), "classifier.1")), p, training, rand)
    c = _dropout(F.relu(lin(c, "classifier.4")), p * 0.5, training, rand)
    return lin(c, "classifier.7")


# ----------------------------------------------------------------------------------------
# a8 InfoNCE (modules/loss.py:54-85)    a11 tag prediction loss (modules/loss.py:107-265)
# ----------------------------------------------------------------------------------------
def infonce(c, t, layer_idx, weight, tau):
    logits = F.normalize(c, p=2, dim=-1) @ F.normalize(t, p=2, dim=-1).T / tau
    ce = F.cross_entropy(logits, torch.arange(c.shape[0]))
    return ce * weight * (1.0 / (layer_idx * 0.5 + 1))


def _focal(cfg, logits, targets, gamma, alpha, grad_mode):
    C = logits.shape[-1]
    onehot = torch.zeros_like(logits).scatter_(1, targets.unsqueeze(1), 1)
    if cfg.use_label_smoothing and grad_mode:  # loss.py:247
        s = min(0.25, cfg.label_smoothing_alpha + gamma * 0.015 + min(0.3, 0.05 * (C / 100)))
        onehot = onehot * (1 - s) + s / C
    pt = (onehot * F.softmax(logits, dim=-1)).sum(1)
    ce = -(onehot * F.log_softmax(logits, dim=-1)).sum(1)
    return (alpha * (1 - pt) ** gamma * ce).mean()


def tag_prediction_loss(cfg, logits, target, rand):
    """Always evaluated with layer_idx=0 and class_counts=None."""
    valid = target >= 0
    if int(valid.sum()) == 0:  # loss.py:123-125
        return torch.tensor(0.0), torch.tensor(0.0)
    vl, vt = logits[valid], target[valid]
    acc = (vl.argmax(-1) == vt).float().mean()
    grad_mode = vl.requires_grad
    mixed = cfg.use_mixup and vl.shape[0] > 1 and grad_mode  # loss.py:139-141
    if mixed:
        perm, lam = rand.mixup(vl.shape[0])
        vl = lam * vl + (1 - lam) * vl[perm]
        ta, tb = vt, vt[perm]
    if cfg.use_focal_loss:
        fp = cfg.focal_loss_params or {"gamma": 2.0, "alpha": 0.25}  # loss.py:93
        gamma = fp.get("gamma_0", fp.get("gamma", 2.0))
        alpha = max(0.08, fp.get("alpha_0", fp.get("alpha", 0.25)))
        if mixed:
            loss = lam * _focal(cfg, vl, ta, gamma, alpha, grad_mode) + (1 - lam) * _focal(cfg, vl, tb, gamma, alpha, grad_mode)
        else:
            loss = _focal(cfg, vl, vt, gamma, alpha, grad_mode)
    else:
        ls = 0.05  # loss.py:205 with layer_idx == 0
        probs = F.softmax(logits[valid], dim=-1)  # loss.py:136 (un-mixed logits)
        if mixed:
            ce = lam * F.cross_entropy(vl, ta, label_smoothing=ls) + (1 - lam) * F.cross_entropy(vl, tb, label_smoothing=ls)
        else:
            ce = F.cross_entropy(vl, vt, label_smoothing=ls)
        uniform = torch.ones_like(probs) / probs.shape[-1]
        kl = F.kl_div(torch.log(probs + 1e-8), uniform, reduction="batchmean") * 0.05
        loss = ce + kl  # the l2_reg loop at loss.py:209-211 never runs (a Tensor has no .parameters)
    return loss, acc


# ----------------------------------------------------------------------------------------
# a12 uniqueness loss exactly as forward() calls it: ids transposed to [L,B]
# ----------------------------------------------------------------------------------------
def uniqueness_as_called(ids_LB, enc, weight, margin):
    n, _ = ids_LB.shape  # n == L: the levels play the role of the batch (h_rqvae.py:52,630)
    if n <= 1:
        return torch.tensor(0.0)
    eq = (ids_LB.unsqueeze(1) == ids_LB.unsqueeze(0)).all(-1)
    a, b = torch.where(eq & ~torch.eye(n, dtype=torch.bool))
    keep = a < b
    a, b = a[keep], b[keep]
    if len(a) == 0:
        return torch.tensor(0.0)
    cos = (F.normalize(enc[a], p=2, dim=-1) * F.normalize(enc[b], p=2, dim=-1)).sum(-1)
    return weight * F.relu(cos - margin).mean()


def p_unique(ids_BL):
    # h_rqvae.py:646-648, the O(B^2 L) form the reference runs (kept for faithful CPU timing)
    eq = (ids_BL.unsqueeze(1) == ids_BL.unsqueeze(0)).all(-1)
    return (~torch.triu(eq, diagonal=1)).all(dim=1).sum() / ids_BL.shape[0]


def p_unique_fast(ids_BL):
    return torch.tensor(float(torch.unique(ids_BL, dim=0).shape[0]) / ids_BL.shape[0])


# ----------------------------------------------------------------------------------------
# a1/a7  the whole forward  (h_rqvae.py:481-583, 585-672)
# ----------------------------------------------------------------------------------------
def forward(P, cfg: Cfg, x, tags_emb=None, tags_indices=None, gumbel_t=0.2, training=True,
            rand=None, bn_buffers=None, quadratic_unique=True):
    rand = rand or TorchRand(cfg.mixup_alpha)
    L = cfg.n_layers
    x = x.float()
    z = mlp(x, enc_weights(P, cfg), cfg.codebook_normalize)
    res = z
    embs, residuals, ids_l, dists = [], [], [], []
    qloss = torch.zeros(())
    tagged = tags_emb is not None and tags_indices is not None
    align_l, pred_l, acc_l = [], [], []
    for i in range(L):
        residuals.append(res)
        cb = effective_codebook(P, cfg, i)
        out, ids, loss, dist = quantize_level(res, cb, cfg.codebook_mode, cfg.commitment_weight,
                                             training, gumbel_t, rand)
        qloss = qloss + loss
        embs.append(out)
        ids_l.append(ids)
        dists.append(dist.detach())
        if tagged:
            concat = torch.cat(embs, dim=-1)
            proj = tag_projector(P, cfg, i, tags_emb[:, i].float(), training, rand, bn_buffers)
            align_l.append(infonce(concat, proj, i, cfg.tag_alignment_weight, cfg.alignment_temperature))
            logits = tag_predictor(P, cfg, i, concat, training, rand)
            pl, pa = tag_prediction_loss(cfg, logits, tags_indices[:, i], rand)
            pred_l.append(pl)
            acc_l.append(pa)
        res = res - out
    emb = torch.stack(embs, dim=-1)  # [B, D, L]
    sem_ids = torch.stack(ids_l, dim=-1)  # [B, L]
    x_hat = mlp(emb.sum(-1), dec_weights(P, cfg), True)
    c = cfg.n_cat_features
    if c == 0:  # (the forward's second l2norm / cat is an identity then)
        recon = ((x_hat - x) ** 2).sum(-1)
    else:  # h_rqvae.py:610-613 + loss.py:15-33
        x_hat = torch.cat([F.normalize(x_hat[..., :-c], p=2, dim=-1, eps=1e-12), x_hat[..., -c:]], dim=-1)
        recon = ((x_hat[:, :-c] - x[:, :-c]) ** 2).sum(-1) \
            + F.binary_cross_entropy_with_logits(x_hat[:, -c:], x[:, -c:], reduction="none").sum(-1)
    if tagged:
        align = torch.stack(align_l).sum() / L
        pred = torch.stack(pred_l).sum() / L
        acc = torch.stack(acc_l).sum() / L
    else:
        align = pred = acc = torch.zeros(())
    uniq = uniqueness_as_called(sem_ids.T, z, cfg.sem_id_uniqueness_weight, cfg.sem_id_uniqueness_margin)
    total = (recon.mean() + qloss.mean() + cfg.tag_alignment_weight * align
             + cfg.tag_prediction_weight * pred + cfg.sem_id_uniqueness_weight * uniq)
    with torch.no_grad():
        embs_norm = emb.norm(dim=1)
        pu = p_unique(sem_ids) if quadratic_unique else p_unique_fast(sem_ids)
    return dict(loss=total, reconstruction_loss=recon, rqvae_loss=qloss, tag_align_loss=align,
                tag_pred_loss=pred, tag_pred_accuracy=acc, embs_norm=embs_norm, p_unique_ids=pu,
                tag_align_loss_by_layer=torch.stack(align_l) if tagged else None,
                tag_pred_loss_by_layer=torch.stack(pred_l) if tagged else None,
                tag_pred_accuracy_by_layer=torch.stack(acc_l) if tagged else None,
                sem_id_uniqueness_loss=uniq, z=z, x_hat=x_hat, embeddings=emb,
                residuals=torch.stack(residuals, dim=-1), sem_ids=sem_ids, dists=dists)


# ----------------------------------------------------------------------------------------
# parameter construction: same shapes / key names as the reference state dict
# ----------------------------------------------------------------------------------------
def param_shapes(cfg: Cfg, with_tags=True):
    dims = [cfg.input_dim] + list(cfg.hidden_dims) + [cfg.embed_dim]
    sh = {}
    for j, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        sh[f"encoder.mlp.{2 * j}.weight"] = (b, a)
    rd = dims[::-1]
    for j, (a, b) in enumerate(zip(rd[:-1], rd[1:])):
        sh[f"decoder.mlp.{2 * j}.weight"] = (b, a)
    for i in range(cfg.n_layers):
        sh[f"layers.{i}.embedding.weight"] = (cfg.codebook_size, cfg.embed_dim)
        if cfg.codebook_sim_vq:
            sh[f"layers.{i}.out_proj.0.weight"] = (cfg.embed_dim, cfg.embed_dim)
    if not with_tags:
        return sh
    H0 = cfg.hidden_dims[0]
    for i, C in enumerate(cfg.classes()):
        e, hidden, mid, _ = predictor_dims(cfg, i)
        p = f"tag_projectors.{i}."
        sh[p + "0.weight"], sh[p + "0.bias"] = (H0, cfg.tag_embed_dim), (H0,)
        if cfg.use_batch_norm:
            sh[p + "1.weight"], sh[p + "1.bias"] = (H0,), (H0,)
        sh[p + "4.weight"], sh[p + "4.bias"] = (e, H0), (e,)
        if cfg.codebook_normalize:
            sh[p + "5.weight"], sh[p + "5.bias"] = (e,), (e,)
        q = f"tag_predictors.{i}."
        for name, (o, n) in {"attention.0": (e // 4, e), "attention.2": (e // 2, e // 4), "attention.4": (e, e // 2),
                             "feature_extractor.0": (hidden, e), "residual_block1.0": (mid, hidden),
                             "residual_block1.4": (hidden, mid), "residual_block2.0": (mid, hidden),
                             "residual_block2.4": (hidden, mid), "classifier.0": (mid, hidden),
                             "classifier.4": (mid // 2, mid), "classifier.7": (C, mid // 2)}.items():
            sh[q + name + ".weight"], sh[q + name + ".bias"] = (o, n), (o,)
        if cfg.use_batch_norm:
            for name, n in {"feature_extractor.1": hidden, "residual_block1.1": mid, "residual_block1.7": hidden,
                            "residual_block2.1": mid, "residual_block2.7": hidden, "classifier.1": mid}.items():
                sh[q + name + ".weight"], sh[q + name + ".bias"] = (n,), (n,)
    return sh


def formula_params(cfg: Cfg, seed=100, with_tags=True):
    """Hash-filled weights at nn.Linear-default scale; codebooks spread so every level uses many
    codes (level 0 random directions, deeper levels shrinking residual-sized vectors)."""
    import numpy as np
    from . import fill
    P = {}
    shapes = param_shapes(cfg, with_tags)
    for n, (k, shp) in enumerate(shapes.items()):
        s = seed + n
        sibling_w = shapes.get(k[: -len("bias")] + "weight", ()) if k.endswith(".bias") else ()
        if k.startswith("layers.") and k.endswith("embedding.weight"):
            lvl = int(k.split(".")[1])
            scale = 1.0 if lvl == 0 else 0.35 * (0.5 ** lvl)
            P[k] = torch.from_numpy(fill.uniform(shp, s, -1.0, 1.0) * np.float32(scale))
        elif len(shp) == 2:
            bound = 1.0 / math.sqrt(shp[1])
            P[k] = torch.from_numpy(fill.uniform(shp, s, -bound, bound))
        elif k.endswith(".weight"):
            P[k] = torch.from_numpy(fill.uniform(shp, s, 0.8, 1.2))  # norm-layer gain
        elif len(sibling_w) == 1:
            P[k] = torch.from_numpy(fill.uniform(shp, s, -0.1, 0.1))  # norm-layer shift
        else:
            P[k] = torch.from_numpy(fill.uniform(shp, s, -0.05, 0.05))  # linear bias
    return P


def formula_batch(cfg: Cfg, B, seed=7, tagged=True, invalid_frac=0.05):
    from . import fill
    import numpy as np
    x = torch.from_numpy(fill.unit_rows((B, cfg.input_dim), seed))
    if cfg.n_cat_features:  # the categorical columns hold 0/1 indicators
        c = cfg.n_cat_features
        x[:, -c:] = (x[:, -c:] > 0).float()
    if not tagged:
        return x, None, None
    te = torch.from_numpy(fill.gauss((B, cfg.n_layers, cfg.tag_embed_dim), seed + 1))
    cols = []
    for i, C in enumerate(cfg.classes()):
        idx = fill.ints((B,), seed + 10 + i, C)
        drop = fill.u01(B, seed + 20 + i) < invalid_frac
        idx = np.where(drop, -1, idx)
        cols.append(idx)
    ti = torch.from_numpy(np.stack(cols, axis=1).astype(np.int64))
    return x, te, ti


def grads(P, cfg, x, te=None, ti=None, **kw):
    """Leaf-parameter gradients of the scalar loss (what loss.backward() leaves in .grad)."""
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items() if v.is_floating_point()}
    out = forward(Pg, cfg, x, te, ti, **kw)
    out["loss"].backward()
    return out, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}


def adamw_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor math (decoupled decay first), step counted from 1."""
    p = p * (1 - lr * wd)
    m = m * b1 + g * (1 - b1)
    v = v * b2 + g * g * (1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p = p - (lr / bc1) * m / (v.sqrt() / math.sqrt(bc2) + eps)
    return p, m, v


def cosine_lr(base_lr, eta_min, t, T_max):
    """Closed form of CosineAnnealingLR after t scheduler steps (train_hidvae.py:636-640)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / T_max)) / 2
