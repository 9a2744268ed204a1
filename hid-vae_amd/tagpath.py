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


class HeadsGradPort:
    """Where d loss / d emb_cat waits when the tag heads ran their own backward EARLY (tag_heads_forward with loss_grad): every level's
    branch runs forward and then straight on into its backward on its own stream, seeded with the loss gradient the caller announced
    (HRqVae.loss_grad_hint), instead of idling until the decoder, the loss launch and the start of loss.backward() have gone by on the
    caller's stream.  The heads read emb_cat through detached leaf views; their gradients stay here until the quantiser's backward
    (ops.RQFn / ops.BottleneckFn) asks for them: collect() joins the level streams into the current one and folds the leaves' gradients
    into one [B, L*D] tensor in ONE launch -- the launch ConcatViewsFn.backward issues on the ordinary path, same operands, same order."""

    def __init__(self, shape):
        self.shape = tuple(shape)
        self.leaves, self.streams, self.done = [], [], []

    def collect(self, as_slices=False):
        """as_slices: -> the list of prefix slices themselves (the quantiser's backward adds them up inside its own launch:
        _C.rq_backward) instead of their sum from a launch of its own"""
        main = torch.cuda.current_stream()
        capturing = torch.cuda.is_current_stream_capturing()
        for k, st in enumerate(self.streams):
            if st is None:
                continue
            if capturing:
                with torch.cuda.stream(st):
                    if not torch.cuda.is_current_stream_capturing():
                        continue  # a graph of its own for this half of the backward: the level streams were joined at the end of the previous one
            if k < len(self.done) and self.done[k] is not None:
                main.wait_event(self.done[k])  # the level's BACKWARD; what the level does after it (its own optimizer update) is not waited for here
            else:
                main.wait_stream(st)
        live = []
        for leaf in self.leaves:
            g, leaf.grad = leaf.grad, None
            if g is not None:
                g = g.contiguous()
                g.record_stream(main)
                live.append(g)
        self.leaves, self.streams, self.done = [], [], []
        flush_layernorm_finals()  # the affine gradients of every LayerNorm of the heads: one launch per step (deferred from their backward)
        if not live:
            return None
        _C.phase_mark("bwd:heads joined (port)")
        if as_slices:
            return live
        return _C.sum_prefix_slices(live, self.shape[0], self.shape[1])


def _loss_seeds(model, g, device):
    """the gradients loss.backward(g) hands the per-level alignment / prediction scalars: g * w_a / L and g * w_p / L in float32, exactly
    what hidvae_loss_bwd computes (ops.StepLossFn.backward); device scalars built once per (g, weights) -- before any graph capture"""
    L = float(model.n_layers)
    key = (float(g), float(model.tag_alignment_weight), float(model.tag_prediction_weight), L, str(device))
    cache = model.__dict__.setdefault("_seed_cache", {})
    if key not in cache:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("tag heads: the loss-gradient seeds must exist before a graph capture (run one eager step first)")
        g32 = np.float32(g)
        cache[key] = (torch.tensor(float(g32 * np.float32(model.tag_alignment_weight / L)), dtype=torch.float32, device=device),
                      torch.tensor(float(g32 * np.float32(model.tag_prediction_weight / L)), dtype=torch.float32, device=device))
    return cache[key]


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


_LN_DEFER = [False]  # True while the heads' early backward is being issued: the finishing launch is left to HeadsGradPort.collect (one per step)
_LN_PENDING = []  # (stream, partials, M, N, gamma Parameter, beta Parameter) of LayerNorm backwards whose affine gradients are not finished yet


def _param_grad_slot(p):
    """-> (dst, accumulate): where a parameter's gradient is written directly -- its flat-buffer slot (ops.grad_sink), else its .grad
    (created here on first use: these gradients do not travel through autograd's AccumulateGrad)"""
    from .ops import grad_sink
    dst, acc = grad_sink(p)
    if dst is not None:
        return dst, acc
    if p.grad is None:
        p.grad = torch.empty_like(p)
        return p.grad, False
    return p.grad, True


def flush_layernorm_finals(only_stream=None):
    """ONE launch finishing the affine gradients (d gamma, d beta) of every LayerNorm backward issued since the last flush
    (hidvae_layernorm_param_final_many).  Queued as an end-of-backward callback by LayerNormFn; runs on the stream backward() was
    called on, after making it wait for the streams the partial sums were produced on.  only_stream: just the LayerNorms whose
    backward ran on that stream (a level's heads finishing their own gradients before their own optimizer update)."""
    global _LN_PENDING
    if only_stream is None:
        pend, _LN_PENDING = _LN_PENDING, []
    else:
        pend = [e for e in _LN_PENDING if e[0].cuda_stream == only_stream.cuda_stream]
        _LN_PENDING = [e for e in _LN_PENDING if e[0].cuda_stream != only_stream.cuda_stream]
    if not pend:
        return
    main = torch.cuda.current_stream()
    seen = set()
    for st, part, *_ in pend:
        if st.cuda_stream != main.cuda_stream and st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            main.wait_stream(st)
        part.record_stream(main)
    # one launch per ROUND: a parameter that several backward nodes contribute to (gradient accumulation: the same LayerNorm in two
    # micro-batches' graphs) is written by one problem per launch, in arrival order -- two problems of one launch adding into the same
    # destination would race
    rounds = []
    seen_in_round = []
    with torch.no_grad():
        for _, part, M, N, gp, bp in pend:
            r = 0
            while r < len(rounds) and id(gp) in seen_in_round[r]:
                r += 1
            if r == len(rounds):
                rounds.append([])
                seen_in_round.append(set())
            seen_in_round[r].add(id(gp))
            rounds[r].append((part, M, N, gp, bp))
        for probs in rounds:
            launch = []
            for part, M, N, gp, bp in probs:
                gdst, gacc = _param_grad_slot(gp)
                bdst, bacc = _param_grad_slot(bp)
                if gacc != bacc:  # (cannot happen for a LayerNorm's own pair; kept exact anyway)
                    (gdst if not gacc else bdst).zero_()
                    gacc = bacc = True
                launch.append((part, M, N, gdst, bdst, gacc))
            _C.layernorm_param_final_many(launch)


class ResidualLink:
    """The gradient of a residual connection f_{n+1} = LN(..) + f_n, carried beside autograd: the LayerNorm that consumes f_n as its
    residual (`res_link`) leaves the gradient of its output here, and the LayerNorm that PRODUCED f_n (`extra_grad`) adds it to the one
    autograd hands it inside its backward launch (gy2 of hidvae_layernorm_bwd_partial).  The residual input itself enters detached, so
    autograd never sees f_n's second consumer and never issues the add."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class LayerNormFn(Function):
    """y = dropout(relu?(LayerNorm(x))) + residual in one launch.
    Backward: ONE launch for gx and the per-4-row partial sums of (d gamma, d beta); the partials of all LayerNorms of a backward pass
    are finished by one launch at its end (flush_layernorm_finals) and written straight into the parameters' gradient slots.  The
    ReLU -> Dropout gate is read off the saved OUTPUT y (> 0 exactly where the unit was active and kept), so no keep-mask is kept for
    the backward.  in_relu_scale != 0: the input x is itself the output of Linear -> ReLU -> Dropout(in_relu_scale) with this
    LayerNorm as its only consumer, and gx is returned already taken through that gate (that Linear is built with act_bwd_done)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, mask, mask_scale, residual, in_relu_scale=0.0, res_link=None, extra_grad=None, computed=None):
        ctx.set_materialize_grads(False)
        if computed is not None:  # (y, mean, rstd) already produced by a fused launch (hidvae_predictor_fwd): nothing is launched here
            y, mean, rstd = computed
        else:
            y, mean, rstd = _C.layernorm_fwd(x, gamma, beta, eps, relu, mask, mask_scale, residual)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, y if relu else None)
        ctx.cfg = (relu, float(mask_scale) if mask is not None else 1.0, residual is not None, float(in_relu_scale))
        ctx.gamma_param, ctx.beta_param = gamma, beta
        ctx.res_link, ctx.extra_grad = res_link, extra_grad  # (ResidualLink: the residual's gradient travels beside autograd)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 12
        x, gamma, beta, mean, rstd, y = ctx.saved_tensors
        relu, scale, has_res, in_relu_scale = ctx.cfg
        gy = gy.contiguous()
        gy2 = None
        if ctx.extra_grad is not None:  # this output also fed a residual connection: its gradient waits in the link
            gy2, ctx.extra_grad.g = ctx.extra_grad.g, None
        link = ctx.res_link
        res_grad = gy if (has_res and link is None) else None  # (with a link the residual input was detached)
        if x.shape[1] > 1024:  # rows too wide for the register-resident kernel: the two-launch form, gradients through autograd
            from .ops import grad_sink
            gdst, gacc = grad_sink(ctx.gamma_param)
            bdst, bacc = grad_sink(ctx.beta_param)
            if gdst is None or bdst is None or gacc != bacc:
                gdst = bdst = None
            if gy2 is not None:
                gy = gy + gy2
            if link is not None:
                link.g = gy
            gate = None if y is None else (y > 0).to(x.dtype)
            gx, gg, gb = _C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, gate, scale, need_gx=ctx.needs_input_grad[0],
                                              gg=gdst, gb=bdst, accumulate=gacc)
            if in_relu_scale != 0.0 and gx is not None:
                gx = _C.act_bwd(gx, x, _C.EPI_RELU, x, in_relu_scale)
            if gdst is not None:
                gg = gb = None
            return gx, gg, gb, None, None, None, None, res_grad, None, None, None, None
        if link is not None and gy2 is not None:
            gx, part, link.g = _C.layernorm_bwd_partial(gy, x, gamma, beta, mean, rstd, relu, y, scale, in_relu_scale,
                                                        need_gx=ctx.needs_input_grad[0], gy2=gy2, want_sum=True)
        else:
            if link is not None:
                link.g = gy
            gx, part = _C.layernorm_bwd_partial(gy, x, gamma, beta, mean, rstd, relu, y, scale, in_relu_scale,
                                                need_gx=ctx.needs_input_grad[0], gy2=gy2)
        if not _LN_PENDING and not _LN_DEFER[0]:
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(flush_layernorm_finals)
        _LN_PENDING.append((torch.cuda.current_stream(), part, x.shape[0], x.shape[1], ctx.gamma_param, ctx.beta_param))
        return gx, None, None, None, None, None, None, res_grad, None, None, None, None


_PARAM_LATER = [None]  # a list while a level's early backward is being issued: launches that only produce PARAMETER gradients wait in it
                       # until the level's gradient hand-over has been signalled (tag_heads_forward flushes it on the same stream)


class GateFn(Function):
    """TagPredictor's attention gate (h_rqvae.py:128-139, :196-206) as one launch each way (_C.gate_fwd / _C.gate_bwd) plus one grouped
    launch for the three Linears' weight and bias gradients.  x: the level's concat-embedding view [B, E] (a column prefix of
    emb_cat); the returned gradient covers BOTH uses of x (attention input and gate input)."""

    @staticmethod
    def forward(ctx, x, W0, b0, W2, b2, W4, b4, normalize):
        ctx.set_materialize_grads(False)
        h, saved = _C.gate_fwd(x, W0, b0, W2, b2, W4, b4, normalize)
        a1, pre2, a2, a3, nrm = saved
        ctx.save_for_backward(x, W0, W2, W4, a1, pre2, a2, a3, nrm)
        ctx.normalize = bool(normalize)
        ctx.params = (W0, b0, W2, b2, W4, b4)
        ctx.need_x = ctx.needs_input_grad[0]
        return h

    @staticmethod
    def backward(ctx, gh):
        if gh is None:
            return (None,) * 8
        from .ops import grad_sink
        x, W0, W2, W4, a1, pre2, a2, a3, nrm = ctx.saved_tensors
        gx, g3, g2, g1 = _C.gate_bwd(gh.contiguous(), x, W0, W2, W4, ctx.normalize, (a1, pre2, a2, a3, nrm))
        probs, sunk = [], []
        later = _PARAM_LATER[0]
        for g, inp, wp, bp in ((g1, x, ctx.params[0], ctx.params[1]), (g2, a1, ctx.params[2], ctx.params[3]), (g3, a2, ctx.params[4], ctx.params[5])):
            if later is not None:
                # deferred (below): the launch writes the parameters' gradient slots / .grad itself and autograd is handed nothing -- a
                # tensor returned now and filled later would be CLONED by AccumulateGrad while still empty (it sees our reference to it)
                dst, acc = _param_grad_slot(wp)
                bdst, bacc = _param_grad_slot(bp)
            else:
                dst, acc = grad_sink(wp)
                bdst, bacc = grad_sink(bp)
            probs.append(dict(g=g, x=inp, w=wp, need_dx=False, dW=dst, accumulate=acc, bias=True, db=bdst, accumulate_db=bacc))
            sunk.append((dst is not None, bdst is not None))
        if later is not None:
            # the three weight / bias gradients are not on the way to emb_cat: their (grouped) launch is issued behind the level's
            # hand-over event (12.8 us of the widest level's lane in front of that event otherwise)
            res = [(pr["dW"], None, pr["db"]) for pr in probs]
            later.append(lambda probs=probs: _C.linear_bwd_group(probs))
        else:
            res = _C.linear_bwd_group(probs)
        out = [gx if ctx.need_x else None]
        for (dW, _, db), (ws, bs) in zip(res, sunk):
            out += [None if ws else dW, None if bs else db]
        return tuple(out) + (None,)


class BatchNormFn(Function):
    """y = dropout(relu?(BatchNorm1d(x))) in one launch; training updates the running statistics in place.  mask: a keep-mask tensor
    (injected draws) or a _C.DropSpec (decided inside the launch); the backward reads the gate off the saved output either way."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, momentum, eps, training, relu, mask, mask_scale):
        ctx.set_materialize_grads(False)
        y, sm, sr = _C.batchnorm_fwd(x, gamma, beta, eps, momentum, training, running_mean, running_var, relu, mask, mask_scale,
                                     num_batches=num_batches)
        gate_from_y = relu  # (Dropout without ReLU does not occur in the reference's heads; a bare mask then stays a tensor)
        if mask is not None and not relu and not torch.is_tensor(mask):
            raise NotImplementedError("in-kernel dropout after a BatchNorm without ReLU")
        ctx.save_for_backward(x, gamma, beta, sm, sr, y if gate_from_y else None, mask if (torch.is_tensor(mask) and not gate_from_y) else None)
        ctx.cfg = (relu, float(mask_scale) if mask is not None else 1.0, training)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 12
        x, gamma, beta, sm, sr, y, mask = ctx.saved_tensors
        relu, mask_scale, training = ctx.cfg
        if not training:
            raise RuntimeError("BatchNorm1d in eval mode is not differentiated on the HIP path")
        gx, gg, gb = _C.batchnorm_bwd(gy.contiguous(), x, gamma, beta, sm, sr, relu, mask, mask_scale, need_gx=ctx.needs_input_grad[0], y_out=y)
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
        cn, nc, tn, nt = _C.l2norm_fwd_pair(c, t)  # both normalisations in one launch
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
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # d cn = dS tn and d tn = dS^T cn are the dX and the dW of one "Linear" whose weight is tn: ONE paired launch; then both
            # normalisations' backward in one launch (5 launches -> 3)
            gtn, gcn = _C.linear_bwd(dS, cn, tn, True)
            gc, gt = _C.l2norm_bwd_pair(gcn, cn, nc, gtn, tn, nt)
            return gc, gt, None, None
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


# ------------------------------------------------------------------------------------------------ compositions
def _lin(x, m, act=_C.EPI_NONE, mask=None, scale=1.0, act_bwd_done=False, dx_gate=None, computed=None):
    return LinearFn.apply(x, m.weight, m.bias, act, mask, scale, act_bwd_done, dx_gate, computed)


def _mask(rand, shape, p, device, training):
    if not training or p == 0.0:
        return None, 1.0
    return rand.dropout_keep(shape, p, device), _keep_scale(p)


def _lin_norm_relu_drop(x, lin, norm, p, rand, training, in_gate=None, extra_grad=None):
    """Linear -> (LayerNorm) -> ReLU -> Dropout as two launches (GEMM+bias, LN+ReLU+mask) or one (GEMM+bias+ReLU+mask).
    -> (y, scale): y = relu(.) * keep * scale.  extra_grad: a ResidualLink whose gradient the LayerNorm's backward adds in (LayerNorm only)."""
    mask, scale = _mask(rand, (x.shape[0], lin.out_features), p, x.device, training)
    if isinstance(norm, nn.LayerNorm):
        return LayerNormFn.apply(_lin(x, lin, dx_gate=in_gate), norm.weight, norm.bias, norm.eps, True, mask, scale, None, 0.0, None,
                                 extra_grad), scale
    return _lin(x, lin, _C.EPI_RELU, mask, scale, dx_gate=in_gate), scale


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


def _predictor_fusable(pred, x, rand):
    """the one-launch forward (hidvae_predictor_fwd) takes a predictor in training with every hop a LayerNorm, dropout decided in the
    launch, and no layer wider than 256"""
    if os.environ.get("HIDVAE_FUSED_PREDICTOR", "1") == "0" or not (pred.training and torch.is_grad_enabled()) or x.dim() != 2:
        return False
    E = x.shape[1]
    if E % 4 != 0 or E > 128 or x.shape[0] < 16:
        return False
    fe, blocks, cl = pred.feature_extractor, (pred.residual_block1, pred.residual_block2), pred.classifier
    if not (isinstance(fe[1], nn.LayerNorm) and isinstance(cl[1], nn.LayerNorm)
            and all(isinstance(rb[1], nn.LayerNorm) and isinstance(rb[7], nn.LayerNorm) for rb in blocks)):
        return False
    lins = [fe[0], cl[0], cl[4], cl[7]] + [rb[k] for rb in blocks for k in (0, 4)]
    if max(max(m.out_features, m.in_features) for m in lins) > _C.PRED_WMAX or any(m.bias is None for m in lins):
        return False
    return pred.dropout_p == 0.0 or hasattr(rand, "state")  # (a provider of mask TENSORS -- the parity tests' injected masks -- keeps the launches apart)


class PredictorFn(Function):
    """Everything of a narrow TagPredictor behind its gate as ONE node: forward = hidvae_predictor_fwd, backward = hidvae_predictor_bwd
    (the whole input-gradient chain) + the eight weight / bias gradients from grouped launches + the LayerNorms' affine partials
    queued for the step's one finishing launch.  inputs: h (the gate's output), the unit list, then every parameter in unit order
    (weight, bias[, gamma, beta]) so autograd knows them; -> logits."""

    @staticmethod
    def forward(ctx, h, units, *params):
        ctx.set_materialize_grads(False)
        outs = _C.predictor_fwd(h, units)
        logits = outs[-1]["lin"]
        ctx.units, ctx.n_params = units, len(params)
        ctx.outs = outs[:-1]  # (the last unit's tensor is this node's OUTPUT: saved below, not held as an attribute -- no reference cycle)
        ctx.save_for_backward(h, logits)
        return logits

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * (2 + ctx.n_params)
        from .ops import grad_sink
        h, logits = ctx.saved_tensors
        units = ctx.units
        outs = ctx.outs + [dict(lin=logits, y=None, mean=None, rstd=None)]
        g_h, res = _C.predictor_bwd(g.contiguous(), units, outs, h.shape[1])
        problems, x_in = [], h
        for u, o, r in zip(units, outs, res):
            lin = u["lin"]
            dst, acc = grad_sink(lin.weight)
            bdst, bacc = grad_sink(lin.bias)
            problems.append(dict(g=r["g_lin"], x=x_in, w=lin.weight, need_dx=False, dW=dst, accumulate=acc, bias=True, db=bdst, accumulate_db=bacc,
                                 _sinks=(dst is not None, bdst is not None)))
            x_in = o["y"] if u.get("norm") is not None else o["lin"]
        # the wide layers' weight + bias gradients one ring-kernel launch each (hidvae_linear_bwd without dX), the narrow ones together in
        # one grouped launch (all eight grouped: 136 us -- the grouped kernel is the per-wave design)
        B = h.shape[0]
        done, small = [None] * len(problems), []
        for i, pr in enumerate(problems):
            n_out, n_in = pr["w"].shape
            if _C.workspace_bytes(_C.WS_LINEAR_BWD_ZEROED, B, n_out, n_in, 1) > 0:
                dW, _, db = _C.linear_bwd(pr["g"], pr["x"], None, False, dW=pr["dW"], accumulate=pr["accumulate"], bias=True, db=pr["db"],
                                          accumulate_db=pr["accumulate_db"])
                done[i] = (dW, None, db)
            else:
                small.append(i)
        for k in range(0, len(small), 6):
            for i, r in zip(small[k:k + 6], _C.linear_bwd_group([problems[i] for i in small[k:k + 6]])):
                done[i] = r
        grads = []
        for u, r, pr, (dW, _, db) in zip(units, res, problems, done):
            grads += [None if pr["_sinks"][0] else dW, None if pr["_sinks"][1] else db]
            norm = u.get("norm")
            if norm is not None:
                if not _LN_PENDING and not _LN_DEFER[0]:
                    from torch.autograd import Variable
                    Variable._execution_engine.queue_callback(flush_layernorm_finals)
                _LN_PENDING.append((torch.cuda.current_stream(), r["partials"], h.shape[0], norm.weight.shape[0], norm.weight, norm.bias))
                grads += [None, None]
        return (g_h, None) + tuple(grads)


def _tag_predictor_forward_fused(pred, x, rand):
    """tag_predictor_forward with everything behind the gate computed by ONE launch; the autograd tape is then laid by the same
    Functions as on the unfused path, each handed its output instead of launching (so the backward is the unfused backward)."""
    training, p = pred.training, pred.dropout_p
    att = pred.attention
    h = GateFn.apply(x, att[0].weight, att[0].bias, att[2].weight, att[2].bias, att[4].weight, att[4].bias, bool(pred.apply_norm))
    fe, blocks, cl = pred.feature_extractor, (pred.residual_block1, pred.residual_block2), pred.classifier
    B, dev = x.shape[0], x.device
    # the dropout sites in the order the unfused path requests them
    m_fe = _mask(rand, (B, fe[0].out_features), p, dev, training)
    m_rb = []
    for rb in blocks:
        m_rb.append((_mask(rand, (B, rb[0].out_features), p, dev, training), _mask(rand, (B, rb[4].out_features), p, dev, training)))
    m_c0 = _mask(rand, (B, cl[0].out_features), p, dev, training)
    m_c4 = _mask(rand, (B, cl[4].out_features), p * 0.5, dev, training)
    units = [dict(lin=fe[0], norm=fe[1], act2=True, drop2=m_fe, carry=True)]
    for rb, (m0, m4) in zip(blocks, m_rb):
        units.append(dict(lin=rb[0], norm=rb[1], act2=True, drop2=m0))
        units.append(dict(lin=rb[4], norm=rb[7], act1=True, drop1=m4, residual=True))
    units += [dict(lin=cl[0], norm=cl[1], act2=True, drop2=m_c0), dict(lin=cl[4], norm=None, act1=True, drop1=m_c4), dict(lin=cl[7], norm=None)]
    # ... and, on request, its backward as one node too (PredictorFn: hidvae_predictor_bwd + the wide layers' weight gradients on the ring
    # kernel).  Off by default: it takes level 0's backward from 26 launches / 342 us of lane time to ~12 / ~190, but once the forward is
    # one launch the caller's lane is no longer the longest -- the step ends with level 2's lane -- and the 64-workgroup kernel costs the
    # other lanes what it saves: 1.053-1.066 vs 1.059-1.061 ms (profiles/r04_fused_predictor_bwd_ab.log; with all weight gradients of the
    # level in ONE launch of a ring kernel generalised to a list of problems: 1.088 vs 1.078 on another box, and 1-2 % on every ring
    # launch for the extra indirection -- not kept)
    if os.environ.get("HIDVAE_FUSED_PREDICTOR_BWD", "0") == "1":
        params = [t for u in units for t in ([u["lin"].weight, u["lin"].bias] + ([u["norm"].weight, u["norm"].bias] if u.get("norm") is not None else []))]
        return PredictorFn.apply(h, units, *params)
    o = _C.predictor_fwd(h.detach(), units)
    # ---- the tape: tag_predictor_forward's chain, node for node
    link = ResidualLink()
    k = 0
    f = LayerNormFn.apply(_lin(h, fe[0], computed=o[k]["lin"]), fe[1].weight, fe[1].bias, fe[1].eps, True, m_fe[0], m_fe[1], None, 0.0, None, link,
                          (o[k]["y"], o[k]["mean"], o[k]["rstd"]))
    k += 1
    for n, (rb, (m0, m4)) in enumerate(zip(blocks, m_rb)):
        r = LayerNormFn.apply(_lin(f, rb[0], computed=o[k]["lin"]), rb[1].weight, rb[1].bias, rb[1].eps, True, m0[0], m0[1], None, 0.0, None, None,
                              (o[k]["y"], o[k]["mean"], o[k]["rstd"]))
        k += 1
        r = _lin(r, rb[4], _C.EPI_RELU, m4[0], m4[1], act_bwd_done=True, computed=o[k]["lin"])
        nxt = ResidualLink() if n + 1 < len(blocks) else None
        f = LayerNormFn.apply(r, rb[7].weight, rb[7].bias, rb[7].eps, False, None, 1.0, f.detach(), m4[1], link, nxt,
                              (o[k]["y"], o[k]["mean"], o[k]["rstd"]))
        link = nxt
        k += 1
    c = LayerNormFn.apply(_lin(f, cl[0], computed=o[k]["lin"]), cl[1].weight, cl[1].bias, cl[1].eps, True, m_c0[0], m_c0[1], None, 0.0, None, None,
                          (o[k]["y"], o[k]["mean"], o[k]["rstd"]))
    k += 1
    c = _lin(c, cl[4], _C.EPI_RELU, m_c4[0], m_c4[1], act_bwd_done=True, computed=o[k]["lin"])
    k += 1
    return _lin(c, cl[7], dx_gate=m_c4[1], computed=o[k]["lin"])


def tag_predictor_forward(pred, x, x_gate=None, rand=None):
    """pred: modules.h_rqvae.TagPredictor; x: [batch, embed_dim] (x_gate: accepted for older callers, the same data as x).
    Launches per call, forward: the gate 1, then GEMM (+ LayerNorm) per layer; backward: every activation / dropout gate rides in the
    launch that produces the gradient it applies to (LayerNorm backward or the next layer's input-gradient epilogue), and the
    LayerNorms' affine gradients are finished by one launch per backward pass."""
    from .rand import default_rand
    rand = rand or pred.rand or default_rand()
    training = pred.training
    if x.dim() != 2:
        raise RuntimeError("TagPredictor expects [batch, embed_dim]")
    if _predictor_fusable(pred, x, rand):
        return _tag_predictor_forward_fused(pred, x, rand)
    att = pred.attention
    E = x.shape[1]
    if E % 4 == 0 and E <= 128:
        h = GateFn.apply(x, att[0].weight, att[0].bias, att[2].weight, att[2].bias, att[4].weight, att[4].bias, bool(pred.apply_norm))
    else:  # widths the row-local gate kernel does not take: the layer-by-layer form
        a = _lin(_lin(_lin(x, att[0], _C.EPI_RELU), att[2], _C.EPI_GELU), att[4], _C.EPI_SIGMOID)
        h = MulFn.apply(x if x_gate is None else x_gate, a)
        if pred.apply_norm:
            h = L2NormFn.apply(h, 1e-12)
    p = pred.dropout_p
    fe = pred.feature_extractor
    blocks = (pred.residual_block1, pred.residual_block2)
    # the residual connections' gradients travel beside autograd when every hop is a LayerNorm (ResidualLink): f_n's producer adds the
    # gradient of f_{n+1} inside its own backward launch instead of autograd adding the two in a launch of its own (6 per step)
    linked = (torch.is_grad_enabled() and isinstance(fe[1], nn.LayerNorm) and all(isinstance(rb[7], nn.LayerNorm) for rb in blocks)
              and max(fe[0].out_features, *(rb[4].out_features for rb in blocks)) <= 1024)
    link = ResidualLink() if linked else None
    f, _ = _lin_norm_relu_drop(h, fe[0], fe[1], p, rand, training, extra_grad=link)
    for n, rb in enumerate(blocks):
        r, _ = _lin_norm_relu_drop(f, rb[0], rb[1], p, rand, training)
        mask, scale = _mask(rand, (r.shape[0], rb[4].out_features), p, r.device, training)
        if isinstance(rb[7], nn.LayerNorm):  # Linear -> ReLU -> Dropout -> LayerNorm (+ f): the LayerNorm backward applies the gate
            r = _lin(r, rb[4], _C.EPI_RELU, mask, scale, act_bwd_done=True)
            nxt = ResidualLink() if (linked and n + 1 < len(blocks)) else None
            f = LayerNormFn.apply(r, rb[7].weight, rb[7].bias, rb[7].eps, False, None, 1.0, f.detach() if linked else f, scale,
                                  link, nxt)  # LN(r) + f in one launch
            link = nxt
        else:
            f = AddFn.apply(f, _lin(r, rb[4], _C.EPI_RELU, mask, scale))
    cl = pred.classifier
    c, _ = _lin_norm_relu_drop(f, cl[0], cl[1], p, rand, training)
    mask, scale = _mask(rand, (c.shape[0], cl[4].out_features), p * 0.5, c.device, training)
    c = _lin(c, cl[4], _C.EPI_RELU, mask, scale, act_bwd_done=True)  # its gate rides in the classifier head's input gradient
    return _lin(c, cl[7], dx_gate=scale)


def tag_prediction_loss(loss_mod, logits, target, layer_idx=0, rand=None, level=None):
    """TagPredictionLoss.forward of the reference (loss.py:107-228); layer_idx only selects focal_params keys."""
    from .rand import default_rand
    rand = rand or loss_mod.rand or default_rand(loss_mod.mixup_alpha)
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


def join_tag_streams(device):
    """the current stream waits for everything queued on the tag-level streams (a graph that ends while the heads' early backward is
    still on its own streams would end with unjoined work)"""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    main = torch.cuda.current_stream()
    for st in _TAG_STREAMS.get(key, [None])[1:]:
        main.wait_stream(st)


def _sites(rand, level, kind):
    """the block of dropout-site numbers of one unit of work (see rand.DeviceRand.at_site); providers without site numbers: no-op"""
    scope = getattr(rand, "at_site", None)
    return scope(32 * level + (16 if kind == "pred" else 0)) if scope is not None else contextlib.nullcontext()


def tag_projectors_early(model, tags_emb, rand):
    """The projectors read only the batch's tag embeddings, not the model's encoder: with every level on a stream of its own they are
    issued at the START of the forward, beside the encoder and the quantiser (78 us during which the level streams had nothing to do),
    instead of after them.  -> {level: projected tags} for the levels that went early, or None (a provider whose draws are numbered in call order -- the parity tests'
    injected masks -- keeps the reference's order: projector i right before InfoNCE i).  Needs early_rand to have issued this step's
    generator advance on the first tag stream."""
    L = model.n_layers
    if not (hasattr(rand, "at_site") and hasattr(rand, "state") and getattr(rand, "_early", False)) or L < 2 or tags_emb.dim() != 3:
        return None
    ev = getattr(rand, "_advanced", None)
    if ev is None:
        return None
    B = tags_emb.shape[0]
    te = tags_emb.reshape(B, -1)
    E = model.tag_embed_dim
    side = _tag_streams(tags_emb.device, L + 1)
    main = torch.cuda.current_stream()
    out = {}
    # (levels 1.. only: the replayed graph runs on three hardware queues, and during the encoder the caller's stream, the first tag
    #  stream -- busy with the step's random draws -- and two projectors are what fits: all three projectors early starve the encoder,
    #  78 -> 147 us, and the step gains nothing (1.201 vs 1.204 ms); levels 1 and 2 early: 1.19 ms)
    for i in range(1, L):
        st = side[i + 1]
        st.wait_stream(main)  # the batch is in place
        if i > 0:
            st.wait_event(ev)  # this step's generator advance (first tag stream)
        tags_emb.record_stream(st)
        with torch.cuda.stream(st), _sites(rand, i, "align"):
            out[i] = tag_projector_forward(model.tag_projectors[i], te[:, i * E:(i + 1) * E], model.training, rand)
    return out


def early_rand(rand, targets, device, n_levels, want_mixup):
    """The step's random draws need nothing from the model: the generator's step counter (one tiny launch) and the mixup pairing (one
    launch; needs only the tag indices) are issued at the START of the forward on the first tag stream, where they run beside the
    encoder instead of in front of it / between the quantiser and the heads.  Only with every level on a stream of its own
    (HIDVAE_TAG_STREAMS=2); tag_heads_forward makes the other levels wait for them.  -> True if the draws were issued here."""
    if os.environ.get("HIDVAE_TAG_STREAMS", "2") != "2" or n_levels < 2 or n_levels > 4 or not hasattr(rand, "state"):
        return False
    rand.state(device)  # (exists before anything is queued on the side stream)
    st = _tag_streams(device, n_levels + 1)
    main = torch.cuda.current_stream()
    st[1].wait_stream(main)
    targets.record_stream(st[1])
    with torch.cuda.stream(st[1]):
        rand.begin_step(device)
        rand._advanced = torch.cuda.Event()  # (what the other levels' early projectors wait for: not the mixup launch behind it)
        rand._advanced.record(st[1])
        made = []
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


def tag_heads_forward(model, emb_cat, tags_emb, tags_indices, defer_join=False, port=None, loss_grad=None, projs=None):
    """-> tuple (A_0..A_{L-1}, P_0..P_{L-1}, acc_0..acc_{L-1}) of 0-d device tensors.
    defer_join: -> (that tuple, join) where join() makes the caller's stream wait for the level branches; the caller issues its own
    work (the decoder) in between, so it runs beside the branches.
    port (HeadsGradPort) + loss_grad (the gradient the caller will hand loss.backward): every level's branch continues into its OWN
    backward right after its forward (see HeadsGradPort); the scalars come back detached."""
    # (round 2 also carried a lockstep form with grouped launches -- 119 launches per step instead of 211 -- which was never faster
    #  than the per-level branches below, 1.96 vs 1.53 ms in round 3: one stream means the sum of the kernel times; it was removed)
    L, D = model.n_layers, model.embed_dim
    rand = model._rand()
    training = model.training
    B = emb_cat.shape[0]
    early_bwd = port is not None and loss_grad is not None and training and torch.is_grad_enabled()
    if early_bwd:
        base = emb_cat.detach()
        views = tuple(base[:, : (i + 1) * D].requires_grad_() for i in range(L) for _ in range(2))  # leaf views: one per consumer
        seed_a, seed_p = _loss_seeds(model, loss_grad, emb_cat.device)
    else:
        views = ConcatViewsFn.apply(emb_cat, L, D, 2)  # two consumers per level: InfoNCE, the predictor (its gate launch covers both uses)
    te = tags_emb.reshape(B, -1)  # [B, L_tags*768]: level i is the column block i (a strided view, no copy)
    ti_rows = tags_indices[:, :L].t().contiguous()  # [L, B]: every level's targets contiguous from ONE copy (not one strided-column copy per level)
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
    if early_bwd:
        mode = "0"  # (the early-backward form deals its own streams below)
    branch = _tag_streams(emb_cat.device, L + (1 if mode == "2" else 0)) if (L > 1 and mode != "0") else None
    if branch is not None and mode == "2":
        branch = [None] + branch[1:]
        lvl_stream = lambda i: branch[i + 1]
    else:
        lvl_stream = lambda i: (branch[i] if branch is not None and i > 0 else None)
    if branch is not None:
        for t in (emb_cat, tags_emb, tags_indices, ti_rows):  # main-stream allocations that the branches (and their backward) read
            for st in branch[1:]:
                t.record_stream(st)
        for st in branch[1:]:
            st.wait_stream(main)
        if early:
            for st in branch[2:]:
                st.wait_stream(branch[1])
    fwd_done = []
    if early_bwd:
        # Levels 1.. on side streams of their own; LEVEL 0 ON THE CALLER'S STREAM, in front of the decoder and the loss.  (Round 3 gave
        # every level a side stream and left the caller's to the decoder and the loss.  In the replayed graph that is four busy branches
        # on three hardware queues -- the count the step replays fastest with, __init__.py -- and the caller's branch was the one left
        # waiting: tools/step_phases.py, profiles/r04_step_phases_tagged_3queues.log: decoder forward done at 947 us of a 1,182 us step,
        # 130 us of decoder / loss / decoder-backward work left behind the heads.  With level 0 on the caller's stream there are three
        # branches and each has a queue: 1.187 -> 1.153 ms.  Not kept: the decoder forward at the head of level 1's stream (1.198 ms: it
        # delays that level); the loss launch on a level stream (hipGraph instantiation segfaults on that topology); four graph queues
        # (1.252 ms: the decoder does run early then, at 189 us, but the loss launch still only gets its queue at 967 us).)
        # Forwards are issued in the reference's order (level by level, projector first: the random draws are numbered in that order),
        # then every level's backward on the same stream.  (Dealing the 2 L units -- projector + alignment, predictor + loss -- over three
        # streams by estimated duration, the caller's included, balanced the streams' end times but not the step: 1.39 vs 1.39 ms.)
        plan = {(k, i): i + 1 for i in range(L) for k in ("pred", "align")}
        if L >= 2 and os.environ.get("HIDVAE_L0_ON_CALLER", "1") != "0":
            plan[("pred", 0)] = plan[("align", 0)] = 0
        side = _tag_streams(emb_cat.device, L + 1)
        lanes = [None] + list(side[1:])
        for st in lanes[1:]:
            for t in (emb_cat, tags_emb, tags_indices, ti_rows):
                t.record_stream(st)
            st.wait_stream(main)
        if early and len(side) > 1:  # this step's generator advance and mixup pairing were issued on the first tag stream (early_rand)
            main.wait_stream(side[1])
            for st in side[2:]:
                st.wait_stream(side[1])
        scal, proj_cut = {}, {}
        split_align = os.environ.get("HIDVAE_ALIGN_SPLIT", "1") != "0"
        for i in range(L):
            for kind in ("align", "pred"):
                st = lanes[plan[(kind, i)]]
                with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                    if kind == "align":
                        _C.phase_mark(f"fwd:level {i} start")
                        if projs is not None and i in projs:
                            proj = projs[i]  # (issued on this stream at the start of the forward: tag_projectors_early)
                        else:
                            with _sites(rand, i, "align"):
                                proj = tag_projector_forward(model.tag_projectors[i], te[:, i * E:(i + 1) * E], training, rand)
                        if split_align and proj.requires_grad:
                            # the projector's backward produces parameter gradients only: cut the tape in front of it, so the level's
                            # gradient hand-over (HeadsGradPort) is signalled BEFORE it is issued
                            cut = proj.detach().requires_grad_()
                            proj_cut[i] = (proj, cut)
                            proj = cut
                        scal[(kind, i)] = (model.tag_alignment_loss(views[2 * i], proj, i),)
                        _C.phase_mark(f"fwd:level {i} projector+infonce done")
                    else:
                        with _sites(rand, i, "pred"):
                            logits = tag_predictor_forward(model.tag_predictors[i], views[2 * i + 1], None, rand)
                        scal[(kind, i)] = tag_prediction_loss(model.tag_prediction_loss, logits, ti_rows[i], 0, rand, level=i)
                        _C.phase_mark(f"fwd:level {i} done")
                if st is not None:
                    for t in scal[(kind, i)]:
                        t.record_stream(main)  # consumed by the total-loss launch on the caller's stream
        for st in lanes[1:]:  # the caller's loss launch waits for the units' FORWARD only
            ev = torch.cuda.Event()
            ev.record(st)
            fwd_done.append((st, ev))
        _LN_DEFER[0] = True  # (the LayerNorms' affine gradients are finished by ONE launch, in HeadsGradPort.collect)
        bwd_done = [None] * L
        level_done = getattr(model, "_level_done_hook", None)  # the training loop's: this level's parameters may take their optimizer update
        try:
            # (not kept: level 0's projector + alignment unit as a guest on another level's lane, behind its host, to shorten the
            #  caller's lane by ~180 us: 1.235-1.273 ms against 1.156 -- profiles/r04_align0_lane_ab.log)
            for i in range(L):
                for kind in ("pred", "align"):
                    st = lanes[plan[(kind, i)]]
                    # (called with the unit's stream current: the engine's end-of-pass synchronisation stays on that stream; the
                    #  caller's stream does not wait for a side stream's backward before HeadsGradPort.collect)
                    with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
                        if kind == "pred" and level_done is not None:
                            # (only where the hand-over is an event of its own -- the per-level hook's case; without it the hand-over
                            #  joins the whole lane and there is nothing to get out of its way)
                            _PARAM_LATER[0] = []  # collected through this level's two units, issued after its hand-over below
                        torch.autograd.backward([scal[(kind, i)][0]], [seed_a if kind == "align" else seed_p])
                        if kind == "align" and st is not None and level_done is not None:
                            # (without the hook -- data parallel, gradient accumulation -- the hand-over joins the whole lane: the
                            #  exchange of the heads' gradients follows it)
                            ev = torch.cuda.Event()  # the gradient hand-over waits for the backward into emb_cat only ...
                            ev.record(st)
                            bwd_done[i] = ev
                        if kind == "align":  # ... then what only produces parameter gradients: the deferred launches,
                            later, _PARAM_LATER[0] = _PARAM_LATER[0], None
                            for fn in later or ():
                                fn()
                        if kind == "align" and i in proj_cut:  # the projector's backward,
                            head, cut = proj_cut.pop(i)
                            g_cut, cut.grad = cut.grad, None
                            if g_cut is not None:
                                torch.autograd.backward([head], [g_cut])
                        if kind == "align" and st is not None and level_done is not None:
                            flush_layernorm_finals(only_stream=st)  # the lane's LayerNorms finish their affine gradients there
                            level_done(i)  # and the level's parameters take their update, beside the rest of the backward
            for i in range(L):
                aligns.append(scal[("align", i)][0].detach())
                preds.append(scal[("pred", i)][0].detach())
                accs.append(scal[("pred", i)][1])
        finally:
            _LN_DEFER[0] = False
            _PARAM_LATER[0] = None
        port.leaves, port.streams, port.done = list(views), lanes[1:], bwd_done
        out = tuple(aligns) + tuple(preds) + tuple(accs)

        def join(target=None):  # target: the stream that is to consume the scalars (default: the caller's)
            for st, ev in fwd_done:
                if target is None:
                    main.wait_event(ev)
                elif st is not target:  # (a stream is ordered behind its own work already; a wait on its own event would be a duplicate edge of the captured graph)
                    target.wait_event(ev)

        if defer_join:
            return out, join
        join()
        return out
    for i in range(L):
        st = lvl_stream(i)
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            c_nce, c_att = views[2 * i], views[2 * i + 1]
            _C.phase_mark(f"fwd:level {i} start")
            with _sites(rand, i, "align"):
                proj = tag_projector_forward(model.tag_projectors[i], te[:, i * E:(i + 1) * E], training, rand)
            align = model.tag_alignment_loss(c_nce, proj, i)
            _C.phase_mark(f"fwd:level {i} projector+infonce done")
            with _sites(rand, i, "pred"):
                logits = tag_predictor_forward(model.tag_predictors[i], c_att, None, rand)
            loss, acc = tag_prediction_loss(model.tag_prediction_loss, logits, ti_rows[i], 0, rand, level=i)
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
