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
        _C.phase_mark("bwd:heads joined (concat views)")
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
        _C.phase_mark(f"bwd:projector batchnorm done (then its first Linear)")
        return (gx, gg, gb) + (None,) * 9


INFONCE_CHUNK_FROM = 4096  # batches from here on never materialise the B x B similarity matrix
INFONCE_CHUNK_COLS = 2048


class InfoNCEFn(Function):
    """scale * CE(normalize(c) normalize(t)^T / tau, arange(B))   (reference loss.py:54-85).
    B < 4096: the similarity matrix S [B,B] is one GEMM and softmax(S/tau) is kept for the backward.  From 4096 items on S is produced
    in chunks of 2048 columns and consumed at once by an online logsumexp (hidvae_infonce_lse_chunk); the backward recomputes each
    chunk.  Peak extra memory B x 2048 floats (164 MB at the 20,000-item k-means warm-up forward, where three levels of saved softmax
    were 4.8 GB)."""

    @staticmethod
    def forward(ctx, c, t, tau, scale):
        ctx.set_materialize_grads(False)
        cn, nc = _C.l2norm_fwd(c)
        tn, nt = _C.l2norm_fwd(t)
        B = cn.shape[0]
        ctx.cfg = (tau, scale)
        if B >= INFONCE_CHUNK_FROM:
            f = lambda: torch.empty((B,), device=cn.device, dtype=torch.float32)
            m, l, diag = f(), f(), f()
            Sc = torch.empty((B, min(B, INFONCE_CHUNK_COLS)), device=cn.device, dtype=torch.float32)
            for col0 in range(0, B, INFONCE_CHUNK_COLS):
                C = min(INFONCE_CHUNK_COLS, B - col0)
                S = _C.gemm(_C.GEMM_NT, cn, tn[col0:col0 + C], out=Sc[:, :C])
                _C.infonce_lse_chunk(S, col0, tau, m, l, diag, col0 == 0)
            loss, lse = _C.infonce_lse_finish(m, l, diag, tau, scale)
            ctx.save_for_backward(cn, nc, tn, nt, lse)
            ctx.chunked = True
            return loss
        S = _C.gemm(_C.GEMM_NT, cn, tn)
        loss = _C.infonce_rows(S, tau, scale)  # S now holds softmax(S / tau)
        ctx.save_for_backward(cn, nc, tn, nt, S)
        ctx.chunked = False
        return loss

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        tau, scale = ctx.cfg
        gc = gt = None
        _C.phase_mark(f"bwd:infonce w={ctx.saved_tensors[0].shape[1]} start")
        if ctx.chunked:
            cn, nc, tn, nt, lse = ctx.saved_tensors
            B, w = cn.shape
            g = g.contiguous()
            want_c, want_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
            gcn = torch.empty((B, w), device=cn.device, dtype=torch.float32) if want_c else None
            gtn = torch.empty((B, w), device=cn.device, dtype=torch.float32) if want_t else None
            Sc = torch.empty((B, min(B, INFONCE_CHUNK_COLS)), device=cn.device, dtype=torch.float32)
            for col0 in range(0, B, INFONCE_CHUNK_COLS):
                C = min(INFONCE_CHUNK_COLS, B - col0)
                S = _C.gemm(_C.GEMM_NT, cn, tn[col0:col0 + C], out=Sc[:, :C])
                dS = _C.infonce_dlogits_chunk(S, col0, tau, scale, lse, g)
                if want_c:  # d cn += dS tn[chunk]   (chunks in ascending order: a fixed summation order)
                    _C.gemm(_C.GEMM_NN, dS, tn[col0:col0 + C], out=gcn, split_k=0, accumulate=col0 > 0)
                if want_t:  # d tn[chunk] = dS^T cn
                    _C.gemm(_C.GEMM_TN, dS, cn, out=gtn[col0:col0 + C], split_k=0)
            if want_c:
                gc = _C.l2norm_bwd(gcn, cn, nc)
            if want_t:
                gt = _C.l2norm_bwd(gtn, tn, nt)
            return gc, gt, None, None
        cn, nc, tn, nt, P = ctx.saved_tensors
        dS = _C.infonce_dlogits(P, tau, scale, g.contiguous())
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
        _C.phase_mark(f"bwd:level C={dmix.shape[1] if dmix is not None else '?'} predictor start")
        return (_C.tag_loss_bwd(dmix, dkl, target, inverse, lam, g.contiguous(), nv),) + (None,) * 9


class GroupLinearFn(Function):
    """The same Linear layer of several heads in ONE launch each way (hidvae_gemm_group / hidvae_linear_bwd_group).
    apply(cfgs, *flat) with cfgs = [(act, keep_scale)] per problem and flat = (x, w, b, keep_mask) per problem -> tuple of y.
    Per problem exactly LinearFn's arithmetic: y = act(x W^T + b) [* mask * scale]; backward g' = g * act'(.) * mask*scale (one
    grouped elementwise launch for the problems that need it), then every dW / dX / db of the group from one grid."""

    @staticmethod
    def forward(ctx, cfgs, *flat):
        ctx.set_materialize_grads(False)
        n = len(cfgs)
        probs, pres = [], []
        for i, (act, scale) in enumerate(cfgs):
            x, w, b, mask = flat[4 * i: 4 * i + 4]
            need = any(ctx.needs_input_grad[1 + 4 * i: 1 + 4 * i + 3])
            pre = None
            if need and act in (_C.EPI_SILU, _C.EPI_GELU):
                pre = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32)
            pres.append(pre)
            probs.append(dict(layout=_C.GEMM_NT, A=x, B=w, bias=b, epilogue=act, aux=pre, mask=mask, mask_scale=scale))
        ys = _C.gemm_group(probs)
        ctx.cfgs = cfgs
        ctx.params = [(flat[4 * i + 1], flat[4 * i + 2]) for i in range(n)]  # (the objects: a flat-gradient slot hangs off the Parameter)
        ctx.need_x = [ctx.needs_input_grad[1 + 4 * i] for i in range(n)]
        saved = []
        for i in range(n):
            x, w, b, mask = flat[4 * i: 4 * i + 4]
            saved += [x, w, pres[i] if pres[i] is not None else ys[i], mask]
        ctx.save_for_backward(*saved)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        from .ops import grad_sink
        n = len(ctx.cfgs)
        saved = ctx.saved_tensors
        live = [i for i in range(n) if gs[i] is not None]
        g = {i: gs[i].contiguous() for i in live}
        todo = [i for i in live if ctx.cfgs[i][0] != _C.EPI_NONE or saved[4 * i + 3] is not None]
        if todo:
            outs = _C.act_bwd_group([(g[i], saved[4 * i + 2], ctx.cfgs[i][0], saved[4 * i + 3], ctx.cfgs[i][1]) for i in todo])
            for i, o in zip(todo, outs):
                g[i] = o
        probs, sinks = [], []
        for i in live:
            x, w = saved[4 * i], saved[4 * i + 1]
            wp, bp = ctx.params[i]
            dst, acc = grad_sink(wp)
            pr = dict(g=g[i], x=x, w=w, need_dx=ctx.need_x[i], dW=dst, accumulate=acc)
            bdst = None
            if bp is not None:
                bdst, bacc = grad_sink(bp)
                pr.update(bias=True, db=bdst, accumulate_db=bacc)
            probs.append(pr)
            sinks.append((dst is not None, bdst is not None))
        res = _C.linear_bwd_group(probs) if probs else []
        out = [None] * (1 + 4 * n)
        for i, (dW, dX, db), (wsunk, bsunk) in zip(live, res, sinks):
            out[1 + 4 * i] = dX
            out[2 + 4 * i] = None if wsunk else dW
            out[3 + 4 * i] = None if (bsunk or ctx.params[i][1] is None) else db
        return tuple(out)


class GroupLayerNormFn(Function):
    """LayerNormFn for several heads at once: y = dropout(relu?(LN(x))) + residual, one launch forward, two backward.
    apply(cfgs, *flat): cfgs = [(eps, relu, mask_scale)], flat = (x, gamma, beta, keep_mask, residual) per problem."""

    @staticmethod
    def forward(ctx, cfgs, *flat):
        ctx.set_materialize_grads(False)
        n = len(cfgs)
        outs = _C.layernorm_fwd_group([(flat[5 * i], flat[5 * i + 1], flat[5 * i + 2], cfgs[i][0], cfgs[i][1], flat[5 * i + 3], cfgs[i][2],
                                        flat[5 * i + 4]) for i in range(n)])
        saved = []
        for i in range(n):
            saved += [flat[5 * i], flat[5 * i + 1], flat[5 * i + 2], outs[i][1], outs[i][2], flat[5 * i + 3]]
        ctx.save_for_backward(*saved)
        ctx.cfgs = cfgs
        ctx.params = [(flat[5 * i + 1], flat[5 * i + 2]) for i in range(n)]
        ctx.has_res = [flat[5 * i + 4] is not None for i in range(n)]
        ctx.need_x = [ctx.needs_input_grad[1 + 5 * i] for i in range(n)]
        return tuple(o[0] for o in outs)

    @staticmethod
    def backward(ctx, *gys):
        from .ops import grad_sink
        n = len(ctx.cfgs)
        saved = ctx.saved_tensors
        live = [i for i in range(n) if gys[i] is not None]
        probs, sunk, gyc = [], [], {}
        for i in live:
            x, gamma, beta, mean, rstd, mask = saved[6 * i: 6 * i + 6]
            gyc[i] = gys[i].contiguous()
            gdst, gacc = grad_sink(ctx.params[i][0])
            bdst, bacc = grad_sink(ctx.params[i][1])
            if gdst is None or bdst is None or gacc != bacc:
                gdst = bdst = None
            probs.append(dict(gy=gyc[i], x=x, gamma=gamma, beta=beta, mean=mean, rstd=rstd, relu=ctx.cfgs[i][1], mask=mask,
                              mask_scale=ctx.cfgs[i][2], need_gx=ctx.need_x[i], gg=gdst, gb=bdst, accumulate=gacc))
            sunk.append(gdst is not None)
        res = _C.layernorm_bwd_all_group(probs) if probs else []
        out = [None] * (1 + 5 * n)
        for i, (gx, gg, gb), sk in zip(live, res, sunk):
            out[1 + 5 * i] = gx
            out[2 + 5 * i] = None if sk else gg
            out[3 + 5 * i] = None if sk else gb
            out[5 + 5 * i] = gyc[i] if ctx.has_res[i] else None
        return tuple(out)


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
        st.append(torch.cuda.Stream(device=device))  # (a high-priority stream for the widest level: 4.4 ms instead of 1.8 under graph replay)
        _C.register_ws_lane(st[-1])  # its Linear backwards run next to the other levels': a workspace of their own
    return st


def early_rand(rand, targets, device, n_levels, want_mixup):
    """The step's random draws need nothing from the model: the keep-masks of every dropout site (one torch.bernoulli, ~30 us) and the
    mixup pairing (one torch.rand + hidvae_mixup_plan, ~28 us; needs only the tag indices) are issued at the START of the forward on
    the first tag stream, where they run beside the encoder instead of in front of it / between the quantiser and the heads.  The
    generator is advanced in the same order as before (masks, then pairing), so the draws are the same numbers.
    Only with every level on a stream of its own (HIDVAE_TAG_STREAMS=2); tag_heads_forward makes the other levels wait for them.
    -> True if the draws were issued here."""
    if os.environ.get("HIDVAE_TAG_STREAMS", "2") != "2" or os.environ.get("HIDVAE_TAG_GROUPED", "0") == "1" or n_levels < 2 or n_levels > 4:
        return False
    st = _tag_streams(device, n_levels + 1)
    main = torch.cuda.current_stream()
    st[1].wait_stream(main)
    targets.record_stream(st[1])
    with torch.cuda.stream(st[1]):
        rand.begin_step(device)
        made = [rand._arena] if getattr(rand, "_arena", None) is not None else []
        if want_mixup:
            rand.prepare_mixup(targets, device)
            made += [t for triple in rand._mix for t in triple if torch.is_tensor(t)]
        for t in made:  # allocated on the first tag stream, read on the others (and by their backward)
            for other in st[2:]:
                t.record_stream(other)
            t.record_stream(main)
    rand._early = True
    rand._mix_early = bool(want_mixup)
    return True


def _glin(xs, mods, act=_C.EPI_NONE, masks=None, scales=None):
    """the same Linear of every head in one launch: xs / mods / masks lists over the heads"""
    n = len(xs)
    masks = masks or [None] * n
    scales = scales or [1.0] * n
    flat = []
    for x, m, k in zip(xs, mods, masks):
        flat += [x, m.weight, m.bias, k]
    return list(GroupLinearFn.apply([(act, sc) for sc in scales], *flat))


def _glin_norm_relu_drop(xs, lins, norms, masks, scales):
    """Linear -> (LayerNorm) -> ReLU -> Dropout of every head: two grouped launches (GEMM+bias, LN+ReLU+mask) or one"""
    if isinstance(norms[0], nn.LayerNorm):
        hs = _glin(xs, lins)
        flat = []
        for h, nm, k in zip(hs, norms, masks):
            flat += [h, nm.weight, nm.bias, k, None]
        return list(GroupLayerNormFn.apply([(nm.eps, True, sc) for nm, sc in zip(norms, scales)], *flat))
    return _glin(xs, lins, _C.EPI_RELU, masks, scales)


def tag_heads_forward_grouped(model, emb_cat, tags_emb, tags_indices):
    """tag_heads_forward with the L levels advanced in LOCKSTEP on one stream: every layer position is ONE grouped launch for all
    levels (hidvae_gemm_group / hidvae_linear_bwd_group / hidvae_layernorm_*_group) instead of one launch per level.  The levels'
    heads are structurally identical chains of different widths (h_rqvae.py:108-227, 322-331), small problems whose launch and
    drain latencies then hide under the widest level's arithmetic.  Random draws are requested level by level in the reference's
    order first (dropout masks in forward order, then the mixup pairing), so injected / replayed randomness is unchanged."""
    L, D = model.n_layers, model.embed_dim
    rand = model._rand()
    training = model.training
    B = emb_cat.shape[0]
    dev = emb_cat.device
    views = ConcatViewsFn.apply(emb_cat, L, D, 3)
    te = tags_emb.reshape(B, -1)
    E = model.tag_embed_dim
    lm = model.tag_prediction_loss
    preds_m, projs_m = list(model.tag_predictors), list(model.tag_projectors)
    if training and torch.is_grad_enabled() and lm.use_mixup and B > 1 and hasattr(rand, "prepare_mixup"):
        rand.prepare_mixup(tags_indices[:, :L], dev)
    # ---- the step's random draws, level by level in the reference's order
    plan = []
    for i in range(L):
        pj, pd = projs_m[i], preds_m[i]
        p = pd.dropout_p
        mid, hid, half = pd.residual_block1[0].out_features, pd.feature_extractor[0].out_features, pd.classifier[4].out_features
        sites = [("proj", pj[0].out_features, pj[3].p), ("fe", hid, p), ("rb1a", mid, p), ("rb1b", hid, p), ("rb2a", mid, p), ("rb2b", hid, p),
                 ("cla", mid, p), ("clb", half, p * 0.5)]
        plan.append({name: _mask(rand, (B, width), pp, dev, training) for name, width, pp in sites})
    mk = lambda name: ([plan[i][name][0] for i in range(L)], [plan[i][name][1] for i in range(L)])
    # ---- projectors (Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear -> LayerNorm)
    tes = [te[:, i * E:(i + 1) * E] for i in range(L)]
    masks, scales = mk("proj")
    if isinstance(projs_m[0][1], nn.BatchNorm1d):
        hs = _glin(tes, [pj[0] for pj in projs_m])
        hs = [BatchNormFn.apply(h, pj[1].weight, pj[1].bias, pj[1].running_mean, pj[1].running_var, pj[1].num_batches_tracked, pj[1].momentum,
                                pj[1].eps, training, True, k, sc) for h, pj, k, sc in zip(hs, projs_m, masks, scales)]
    else:
        hs = _glin(tes, [pj[0] for pj in projs_m], _C.EPI_RELU, masks, scales)
    hs = _glin(hs, [pj[4] for pj in projs_m])
    if isinstance(projs_m[0][5], nn.LayerNorm):
        flat = []
        for h, pj in zip(hs, projs_m):
            flat += [h, pj[5].weight, pj[5].bias, None, None]
        hs = list(GroupLayerNormFn.apply([(pj[5].eps, False, 1.0) for pj in projs_m], *flat))
    aligns = [model.tag_alignment_loss(views[3 * i], hs[i], i) for i in range(L)]
    # ---- predictors
    for pd in preds_m:
        if views[0].dim() != 2:
            raise RuntimeError("TagPredictor expects [batch, embed_dim]")
    c_att = [views[3 * i + 1] for i in range(L)]
    a = _glin(c_att, [pd.attention[0] for pd in preds_m], _C.EPI_RELU)
    a = _glin(a, [pd.attention[2] for pd in preds_m], _C.EPI_GELU)
    a = _glin(a, [pd.attention[4] for pd in preds_m], _C.EPI_SIGMOID)
    hh = [MulFn.apply(views[3 * i + 2], a[i]) for i in range(L)]
    hh = [L2NormFn.apply(h, 1e-12) if pd.apply_norm else h for h, pd in zip(hh, preds_m)]
    f = _glin_norm_relu_drop(hh, [pd.feature_extractor[0] for pd in preds_m], [pd.feature_extractor[1] for pd in preds_m], *mk("fe"))
    for rb_name, ka, kb in (("residual_block1", "rb1a", "rb1b"), ("residual_block2", "rb2a", "rb2b")):
        rbs = [getattr(pd, rb_name) for pd in preds_m]
        r = _glin_norm_relu_drop(f, [rb[0] for rb in rbs], [rb[1] for rb in rbs], *mk(ka))
        r = _glin(r, [rb[4] for rb in rbs], _C.EPI_RELU, *mk(kb))
        if isinstance(rbs[0][7], nn.LayerNorm):
            flat = []
            for ri, rb, fi in zip(r, rbs, f):
                flat += [ri, rb[7].weight, rb[7].bias, None, fi]  # LN(r) + f in one launch
            f = list(GroupLayerNormFn.apply([(rb[7].eps, False, 1.0) for rb in rbs], *flat))
        else:
            f = [AddFn.apply(fi, ri) for fi, ri in zip(f, r)]
    cls = [pd.classifier for pd in preds_m]
    c = _glin_norm_relu_drop(f, [cl[0] for cl in cls], [cl[1] for cl in cls], *mk("cla"))
    c = _glin(c, [cl[4] for cl in cls], _C.EPI_RELU, *mk("clb"))
    logits = _glin(c, [cl[7] for cl in cls])
    preds, accs = [], []
    for i in range(L):
        loss, acc = tag_prediction_loss(lm, logits[i], tags_indices[:, i].contiguous(), 0, rand, level=i)
        preds.append(loss)
        accs.append(acc)
    return tuple(aligns) + tuple(preds) + tuple(accs)


def tag_heads_forward(model, emb_cat, tags_emb, tags_indices, defer_join=False):
    """-> tuple (A_0..A_{L-1}, P_0..P_{L-1}, acc_0..acc_{L-1}) of 0-d device tensors.
    defer_join: -> (that tuple, join) where join() makes the caller's stream wait for the level branches; the caller issues its own
    work (the decoder) in between, so it runs beside the branches."""
    # HIDVAE_TAG_GROUPED=1: the lockstep form with grouped launches (119 launches per amazon-shaped step instead of 211).  Measured on
    # MI355X it is NOT faster than the per-level branches below (B=1024: 2.00 vs 1.97 ms; B=2048: 3.24 vs 3.16 ms): these GEMMs are
    # bound by L2->CU operand traffic (4.5-6 TB/s in every variant), not by launch latency, so it stays opt-in.
    if os.environ.get("HIDVAE_TAG_GROUPED", "0") == "1" and 1 < model.n_layers <= 4:
        out = tag_heads_forward_grouped(model, emb_cat, tags_emb, tags_indices)
        return (out, (lambda: None)) if defer_join else out
    L, D = model.n_layers, model.embed_dim
    rand = model._rand()
    training = model.training
    B = emb_cat.shape[0]
    views = ConcatViewsFn.apply(emb_cat, L, D, 3)  # three consumers per level: InfoNCE, attention input, gate input
    te = tags_emb.reshape(B, -1)  # [B, L_tags*768]: level i is the column block i (a strided view, no copy)
    E = model.tag_embed_dim
    aligns, preds, accs = [], [], []
    lm = model.tag_prediction_loss
    early = bool(getattr(rand, "_early", False))       # (the draws of this step are already on their way on the first tag stream:
    mix_early = bool(getattr(rand, "_mix_early", False))  #  early_rand)
    if early:
        rand._early = rand._mix_early = False
    if mix_early:
        pass
    elif training and torch.is_grad_enabled() and lm.use_mixup and B > 1 and hasattr(rand, "prepare_mixup"):
        rand.prepare_mixup(tags_indices[:, :L], emb_cat.device)  # the pairings of all levels from one batched set of launches
    # The L levels' heads are independent of each other (~100 launches each, most of them small), so levels 1..L-1 run on their
    # own streams beside level 0.  Autograd replays each branch's backward on the stream its forward ran on and joins them
    # itself; under HIP-graph capture the fork/join become graph edges (two per extra level and direction -- unlike the per-op
    # forks tried earlier (ops.side_stream) they are amortised over a whole branch).  Measured on the amazon-shaped tagged step:
    # 2.31 ms against 2.64 ms on one stream.  HIDVAE_TAG_STREAMS=0 keeps everything on the caller's stream.
    # (one stream per level: 2.31 ms; additionally splitting projector+alignment from predictor+loss, 2L branches: 2.68 ms)
    main = torch.cuda.current_stream()
    # HIDVAE_TAG_STREAMS: 2 (default) = EVERY level on a stream of its own, the caller's stream is left to the decoder and the loss
    # (1.78-1.79 ms against 1.85-1.86 ms with level 0 on the caller's stream, B = 2048: 2.84 vs 2.91 ms); 1 = level 0 stays with the caller
    mode = os.environ.get("HIDVAE_TAG_STREAMS", "2")
    branch = _tag_streams(emb_cat.device, L + (1 if mode == "2" else 0)) if (L > 1 and mode != "0") else None
    if branch is not None and mode == "2":
        branch = [None] + branch[1:]
        lvl_stream = lambda i: branch[i + 1]
    else:
        lvl_stream = lambda i: (branch[i] if branch is not None and i > 0 else None)
    if branch is not None:
        for t in (emb_cat, tags_emb, tags_indices):  # main-stream allocations that the branches (and their backward) read
            for st in branch[1:]:
                t.record_stream(st)
        for st in branch[1:]:
            st.wait_stream(main)
        if early:
            for st in branch[2:]:
                st.wait_stream(branch[1])
    for i in range(L):
        st = lvl_stream(i)
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            c_nce, c_att, c_gate = views[3 * i], views[3 * i + 1], views[3 * i + 2]
            _C.phase_mark(f"fwd:level {i} start")
            proj = tag_projector_forward(model.tag_projectors[i], te[:, i * E:(i + 1) * E], training, rand)
            align = model.tag_alignment_loss(c_nce, proj, i)
            _C.phase_mark(f"fwd:level {i} projector+infonce done")
            logits = tag_predictor_forward(model.tag_predictors[i], c_att, c_gate, rand)
            loss, acc = tag_prediction_loss(model.tag_prediction_loss, logits, tags_indices[:, i].contiguous(), 0, rand, level=i)
            _C.phase_mark(f"fwd:level {i} done")
        if st is not None:
            for t in (align, loss, acc):
                t.record_stream(main)  # consumed by the total-loss launch on the caller's stream
        aligns.append(align)
        preds.append(loss)
        accs.append(acc)
    out = tuple(aligns) + tuple(preds) + tuple(accs)

    def join():
        if branch is not None:
            for st in branch[1:]:
                main.wait_stream(st)

    if defer_join:
        return out, join
    join()
    return out
