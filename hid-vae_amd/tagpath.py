"""Tag heads of HiD-VAE on the HIP kernels: per level i (reference h_rqvae.py:526-549)
    concat_i = emb_cat[:, :(i+1)*32]                      (a column-prefix VIEW of the RQ kernel's output, no copy)
    proj_i   = tag_projectors[i](tags_emb[:, i])          Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear -> LayerNorm
    A_i      = InfoNCE(concat_i, proj_i)                  loss.py:54-85
    logits_i = tag_predictors[i](concat_i)                h_rqvae.py:196-227
    P_i, acc_i = TagPredictionLoss(logits_i, tags_indices[:, i])     loss.py:107-265 (layer_idx 0)
Every op is a torch.autograd.Function whose forward/backward are C-ABI launches; dropout keep-masks and the mixup
pairing come from the model's randomness provider (hidvae_amd.rand) as device tensors."""
import contextlib
import os

import numpy as np
import torch
from torch import nn
from torch.autograd import Function

from . import _C
from .ops import L2NormFn, LinearFn


def _keep_scale(p):
    # nn.Dropout multiplies by bernoulli(1-p) / (1-p), the division done in float32 (ATen native dropout)
    return float(np.float32(1.0) / np.float32(1.0 - p))


class ConcatViewsFn(Function):
    """emb_cat [B, L*D] -> for every level i, `fan` identical views emb_cat[:, :(i+1)*D] (one per consumer, so autograd
    never has to add fan-out gradients itself); backward folds all of them into one [B, L*D] gradient in ONE launch."""

    @staticmethod
    def forward(ctx, emb_cat, L, D, fan):
        ctx.set_materialize_grads(False)
        ctx.shape = emb_cat.shape
        return tuple(emb_cat[:, : (i + 1) * D] for i in range(L) for _ in range(fan))

    @staticmethod
    def backward(ctx, *grads):
        live = [g.contiguous() for g in grads if g is not None]
        if not live:
            return None, None, None, None
        return _C.sum_prefix_slices(live, ctx.shape[0], ctx.shape[1]), None, None, None


class MulFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a, b)
        return _C.mul(a, b)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        a, b = ctx.saved_tensors
        g = g.contiguous()
        return _C.mul(g, b), _C.mul(g, a)


class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return _C.binary(1, a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class LayerNormFn(Function):
    """y = dropout(relu?(LayerNorm(x))) + residual in one launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, mask, mask_scale, residual):
        ctx.set_materialize_grads(False)
        y, mean, rstd = _C.layernorm_fwd(x, gamma, beta, eps, relu, mask, mask_scale, residual)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, mask)
        ctx.cfg = (relu, mask_scale, residual is not None)
        ctx.gamma_param, ctx.beta_param = gamma, beta
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 8
        x, gamma, beta, mean, rstd, mask = ctx.saved_tensors
        relu, mask_scale, has_res = ctx.cfg
        gy = gy.contiguous()
        from .ops import grad_sink
        gdst, gacc = grad_sink(ctx.gamma_param)
        bdst, bacc = grad_sink(ctx.beta_param)
        if gdst is None or bdst is None or gacc != bacc:
            gdst = bdst = None
        gx, gg, gb = _C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, mask_scale, need_gx=ctx.needs_input_grad[0],
                                          gg=gdst, gb=bdst, accumulate=gacc)
        if gdst is not None:
            gg = gb = None
        return gx, gg, gb, None, None, None, None, (gy if has_res else None)


class BatchNormFn(Function):
    """y = dropout(relu?(BatchNorm1d(x))) in one launch; training updates the running statistics in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, momentum, eps, training, relu, mask, mask_scale):
        ctx.set_materialize_grads(False)
        y, sm, sr = _C.batchnorm_fwd(x, gamma, beta, eps, momentum, training, running_mean, running_var, relu, mask, mask_scale,
                                     num_batches=num_batches)
        ctx.save_for_backward(x, gamma, beta, sm, sr, mask)
        ctx.cfg = (relu, mask_scale, training)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 12
        x, gamma, beta, sm, sr, mask = ctx.saved_tensors
        relu, mask_scale, training = ctx.cfg
        if not training:
            raise RuntimeError("BatchNorm1d in eval mode is not differentiated on the HIP path")
        gx, gg, gb = _C.batchnorm_bwd(gy.contiguous(), x, gamma, beta, sm, sr, relu, mask, mask_scale, need_gx=ctx.needs_input_grad[0])
        return (gx, gg, gb) + (None,) * 9


class InfoNCEFn(Function):
    """scale * CE(normalize(c) normalize(t)^T / tau, arange(B))   (reference loss.py:54-85)."""

    @staticmethod
    def forward(ctx, c, t, tau, scale):
        ctx.set_materialize_grads(False)
        cn, nc = _C.l2norm_fwd(c)
        tn, nt = _C.l2norm_fwd(t)
        S = _C.gemm(_C.GEMM_NT, cn, tn)
        loss = _C.infonce_rows(S, tau, scale)  # S now holds softmax(S / tau)
        ctx.save_for_backward(cn, nc, tn, nt, S)
        ctx.cfg = (tau, scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        cn, nc, tn, nt, P = ctx.saved_tensors
        tau, scale = ctx.cfg
        dS = _C.infonce_dlogits(P, tau, scale, g.contiguous())
        gc = gt = None
        if ctx.needs_input_grad[0]:
            gc = _C.l2norm_bwd(_C.gemm(_C.GEMM_NN, dS, tn, split_k=0), cn, nc)
        if ctx.needs_input_grad[1]:
            gt = _C.l2norm_bwd(_C.gemm(_C.GEMM_TN, dS, cn, split_k=0), tn, nt)
        return gc, gt, None, None


class TagPredLossFn(Function):
    @staticmethod
    def forward(ctx, logits, target, partner, inverse, lam, focal, gamma, alpha, smooth, ce_ls):
        ctx.set_materialize_grads(False)
        want = ctx.needs_input_grad[0]
        loss, acc, nv, dmix, dkl = _C.tag_loss_fwd(logits, target, partner, lam, focal, gamma, alpha, smooth, ce_ls, want)
        ctx.stash = (dmix, dkl, target, inverse, lam, nv)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, _g_acc):
        if g is None:
            return (None,) * 10
        dmix, dkl, target, inverse, lam, nv = ctx.stash
        return (_C.tag_loss_bwd(dmix, dkl, target, inverse, lam, g.contiguous(), nv),) + (None,) * 9


# ------------------------------------------------------------------------------------------------ compositions
def _lin(x, m, act=_C.EPI_NONE, mask=None, scale=1.0):
    return LinearFn.apply(x, m.weight, m.bias, act, mask, scale)


def _mask(rand, shape, p, device, training):
    if not training or p == 0.0:
        return None, 1.0
    return rand.dropout_keep(shape, p, device), _keep_scale(p)


def _lin_norm_relu_drop(x, lin, norm, p, rand, training):
    """Linear -> (LayerNorm) -> ReLU -> Dropout as two launches (GEMM+bias, LN+ReLU+mask) or one (GEMM+bias+ReLU+mask)."""
    mask, scale = _mask(rand, (x.shape[0], lin.out_features), p, x.device, training)
    if isinstance(norm, nn.LayerNorm):
        return LayerNormFn.apply(_lin(x, lin), norm.weight, norm.bias, norm.eps, True, mask, scale, None)
    return _lin(x, lin, _C.EPI_RELU, mask, scale)


def tag_projector_forward(seq, t, training, rand):
    lin0, bn, drop, lin4, ln = seq[0], seq[1], seq[3], seq[4], seq[5]
    mask, scale = _mask(rand, (t.shape[0], lin0.out_features), drop.p, t.device, training)
    if isinstance(bn, nn.BatchNorm1d):
        h = BatchNormFn.apply(_lin(t, lin0), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                              bn.momentum, bn.eps, training, True, mask, scale)
    else:
        h = _lin(t, lin0, _C.EPI_RELU, mask, scale)
    h = _lin(h, lin4)
    if isinstance(ln, nn.LayerNorm):
        h = LayerNormFn.apply(h, ln.weight, ln.bias, ln.eps, False, None, 1.0, None)
    return h


def tag_predictor_forward(pred, x, x_gate=None, rand=None):
    """pred: modules.h_rqvae.TagPredictor; x (and x_gate, the same data as a second autograd leaf of the fan-out)."""
    from .rand import DeviceRand
    rand = rand or pred.rand or DeviceRand()
    training = pred.training
    if x.dim() != 2:
        raise RuntimeError("TagPredictor expects [batch, embed_dim]")
    x_gate = x if x_gate is None else x_gate
    att = pred.attention
    a = _lin(_lin(_lin(x, att[0], _C.EPI_RELU), att[2], _C.EPI_GELU), att[4], _C.EPI_SIGMOID)
    h = MulFn.apply(x_gate, a)
    if pred.apply_norm:
        h = L2NormFn.apply(h, 1e-12)
    p = pred.dropout_p
    fe = pred.feature_extractor
    f = _lin_norm_relu_drop(h, fe[0], fe[1], p, rand, training)
    for rb in (pred.residual_block1, pred.residual_block2):
        r = _lin_norm_relu_drop(f, rb[0], rb[1], p, rand, training)
        mask, scale = _mask(rand, (r.shape[0], rb[4].out_features), p, r.device, training)
        r = _lin(r, rb[4], _C.EPI_RELU, mask, scale)
        if isinstance(rb[7], nn.LayerNorm):
            f = LayerNormFn.apply(r, rb[7].weight, rb[7].bias, rb[7].eps, False, None, 1.0, f)  # LN(r) + f in one launch
        else:
            f = AddFn.apply(f, r)
    cl = pred.classifier
    c = _lin_norm_relu_drop(f, cl[0], cl[1], p, rand, training)
    mask, scale = _mask(rand, (c.shape[0], cl[4].out_features), p * 0.5, c.device, training)
    c = _lin(c, cl[4], _C.EPI_RELU, mask, scale)
    return _lin(c, cl[7])


def tag_prediction_loss(loss_mod, logits, target, layer_idx=0, rand=None, level=None):
    """TagPredictionLoss.forward of the reference (loss.py:107-228); layer_idx only selects focal_params keys."""
    from .rand import DeviceRand
    rand = rand or loss_mod.rand or DeviceRand(loss_mod.mixup_alpha)
    grad_mode = logits.requires_grad and torch.is_grad_enabled()
    C = logits.shape[1]
    fp = loss_mod.focal_params
    gamma = fp.get(f"gamma_{layer_idx}", fp.get("gamma", 2.0)) * (1 + 0.35 * layer_idx)
    alpha = max(0.08, fp.get(f"alpha_{layer_idx}", fp.get("alpha", 0.25)) - 0.06 * layer_idx)
    smooth = 0.0
    if loss_mod.use_label_smoothing and grad_mode:  # loss.py:247-251
        smooth = min(0.25, loss_mod.label_smoothing_alpha + gamma * 0.015 + min(0.3, 0.05 * (C / 100)))
    partner = inverse = lam = None
    if loss_mod.use_mixup and grad_mode and logits.shape[0] > 1:  # loss.py:139-147
        partner, inverse, lam = rand.mixup_partner(target, logits.device, level)
    ce_ls = min(0.25, 0.05 + layer_idx * 0.06)
    return TagPredLossFn.apply(logits.contiguous(), target.contiguous(), partner, inverse, lam, bool(loss_mod.use_focal_loss),
                               float(gamma), float(alpha), float(smooth), float(ce_ls))


_TAG_STREAMS = {}


def _tag_streams(device, n):
    """[None, stream_1, ..., stream_{n-1}] per device (level 0 stays on the caller's stream)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _TAG_STREAMS.setdefault(key, [None])
    while len(st) < n:
        st.append(torch.cuda.Stream(device=device))
    return st


def tag_heads_forward(model, emb_cat, tags_emb, tags_indices):
    """-> tuple (A_0..A_{L-1}, P_0..P_{L-1}, acc_0..acc_{L-1}) of 0-d device tensors."""
    L, D = model.n_layers, model.embed_dim
    rand = model._rand()
    training = model.training
    B = emb_cat.shape[0]
    views = ConcatViewsFn.apply(emb_cat, L, D, 3)  # three consumers per level: InfoNCE, attention input, gate input
    te = tags_emb.reshape(B, -1)  # [B, L_tags*768]: level i is the column block i (a strided view, no copy)
    E = model.tag_embed_dim
    aligns, preds, accs = [], [], []
    lm = model.tag_prediction_loss
    if training and torch.is_grad_enabled() and lm.use_mixup and B > 1 and hasattr(rand, "prepare_mixup"):
        rand.prepare_mixup(tags_indices[:, :L], emb_cat.device)  # the pairings of all levels from one batched set of launches
    # The L levels' heads are independent of each other (~100 launches each, most of them small), so levels 1..L-1 run on their
    # own streams beside level 0.  Autograd replays each branch's backward on the stream its forward ran on and joins them
    # itself; under HIP-graph capture the fork/join become graph edges (two per extra level and direction -- unlike the per-op
    # forks tried earlier (ops.side_stream) they are amortised over a whole branch).  Measured on the amazon-shaped tagged step:
    # 2.31 ms against 2.64 ms on one stream.  HIDVAE_TAG_STREAMS=0 keeps everything on the caller's stream.
    # (one stream per level: 2.31 ms; additionally splitting projector+alignment from predictor+loss, 2L branches: 2.68 ms)
    main = torch.cuda.current_stream()
    branch = _tag_streams(emb_cat.device, L) if (L > 1 and os.environ.get("HIDVAE_TAG_STREAMS", "1") != "0") else None
    if branch is not None:
        for t in (emb_cat, tags_emb, tags_indices):  # main-stream allocations that the branches (and their backward) read
            for st in branch[1:]:
                t.record_stream(st)
        for st in branch[1:]:
            st.wait_stream(main)
    for i in range(L):
        st = branch[i] if branch is not None and i > 0 else None
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            c_nce, c_att, c_gate = views[3 * i], views[3 * i + 1], views[3 * i + 2]
            proj = tag_projector_forward(model.tag_projectors[i], te[:, i * E:(i + 1) * E], training, rand)
            align = model.tag_alignment_loss(c_nce, proj, i)
            logits = tag_predictor_forward(model.tag_predictors[i], c_att, c_gate, rand)
            loss, acc = tag_prediction_loss(model.tag_prediction_loss, logits, tags_indices[:, i].contiguous(), 0, rand, level=i)
        if st is not None:
            for t in (align, loss, acc):
                t.record_stream(main)  # consumed by the total-loss launch on the caller's stream
        aligns.append(align)
        preds.append(loss)
        accs.append(acc)
    if branch is not None:
        for st in branch[1:]:
            main.wait_stream(st)
    return tuple(aligns) + tuple(preds) + tuple(accs)
