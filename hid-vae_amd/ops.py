"""Autograd operators of the tokenizer step: each forward/backward is one or a few C-ABI launches (_C.py).

No torch arithmetic runs on the hot path: torch supplies device memory, streams and the autograd tape only."""
import os

import torch
from torch.autograd import Function

from . import _C


_SIDE = {}


def side_stream():
    """One helper stream per device.  Independent launches (weight gradients next to the input-gradient chain, codebook
    preparation next to the encoder) go there so their fixed per-kernel latencies overlap; under HIP-graph capture the
    fork/join becomes graph edges."""
    if os.environ.get("HIDVAE_SIDE_STREAM", "0") != "1":
        # DEFAULT: everything on the caller's stream.  Measured on MI355X / ROCm 7.2 (bench.py, B=1024, HIP graph): forking the
        # parameter-gradient GEMMs, codebook preparation and id statistics to a second stream made the step SLOWER (0.385 ms
        # vs 0.315 ms untagged, 4.05 ms vs 3.45 ms tagged): every cross-stream edge of the captured graph costs ~5-10 us,
        # more than the ~5 us launches it overlaps.  HIDVAE_SIDE_STREAM=1 re-enables the fork/join structure.
        return torch.cuda.current_stream()
    d = torch.cuda.current_device()
    if d not in _SIDE:
        _SIDE[d] = torch.cuda.Stream()
        _C.register_ws_lane(_SIDE[d])
    return _SIDE[d]


def join_side():
    """Make the current stream wait for everything queued on the helper stream."""
    d = torch.cuda.current_device()
    if d in _SIDE:
        torch.cuda.current_stream().wait_stream(_SIDE[d])


def _join_after_backward():
    """Ask autograd to join the helper stream when the running backward pass ends (so .grad is safe to read on the
    current stream afterwards, whoever reads it)."""
    from torch.autograd import Variable
    Variable._execution_engine.queue_callback(join_side)


def grad_sink(p):
    """-> (dst, accumulate).  dst is parameter p's slot in a flat gradient buffer (parallel.FlatGradBuffer) that the calling
    backward may write in place from its own launch, or None when p has no slot (then the gradient goes back to autograd)."""
    v = getattr(p, "_hv_view", None)
    if v is None or not torch.is_tensor(v):
        return None, False
    acc = bool(p._hv_written)
    p._hv_written = True
    p._hv_nwrites = getattr(p, "_hv_nwrites", 0) + 1  # (FlatGradBuffer.check_first_bucket_untouched reads it)
    return v, acc


def mlp_body_forward(x, weights, keep_for_backward):
    """x W0^T -> SiLU -> ... -> W_last^T  (modules/encoder.py:23-31 of the reference, no biases).
    Returns y and, for the backward, the per-layer (input, pre-activation) tensors."""
    saved = []
    h = x
    last = len(weights) - 1
    for j, w in enumerate(weights):
        if j == last:
            out = _C.gemm(_C.GEMM_NT, h, w)
            saved.append((h, None))
        else:
            pre = torch.empty((h.shape[0], w.shape[0]), device=h.device, dtype=torch.float32) if keep_for_backward else None
            out = _C.gemm(_C.GEMM_NT, h, w, epilogue=_C.EPI_SILU, aux=pre)
            saved.append((h, pre))
        h = out
    return h, saved


def mlp_body_backward(saved, weights, g_y, need_input_grad, in_pre=None):
    """One launch per layer: dW_j = g_j^T h_{j-1} and g_{j-1} = (g_j W_j) * silu'(pre_{j-1}) share a grid (hidvae_linear_bwd).
    in_pre: the stack's input was silu(in_pre) and the gradient wanted is the one w.r.t. in_pre (MLPBackFn)."""
    grads = [None] * len(weights)
    g = g_y
    for j in range(len(weights) - 1, -1, -1):
        inp, _ = saved[j]
        dst, acc = grad_sink(weights[j])
        if j > 0:
            gw, g = _C.linear_bwd(g, inp, weights[j], True, _C.EPI_DSILU, saved[j - 1][1], dW=dst, accumulate=acc)
        elif in_pre is not None and need_input_grad:
            gw, g = _C.linear_bwd(g, inp, weights[j], True, _C.EPI_DSILU, in_pre, dW=dst, accumulate=acc)
        else:
            gw, g = _C.linear_bwd(g, inp, weights[j], need_input_grad, dW=dst, accumulate=acc)
        grads[j] = None if dst is not None else gw  # written in place: autograd has nothing to add
    return g, grads, []


class MLPBodyFn(Function):
    @staticmethod
    def forward(ctx, x, *weights):
        ctx.set_materialize_grads(False)
        y, saved = mlp_body_forward(x, weights, any(ctx.needs_input_grad))
        ctx.saved = saved
        ctx.weights = weights
        ctx.need_x = ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, g_y):
        if g_y is None:
            return (None,) * (1 + len(ctx.weights))
        gx, gws, keep = mlp_body_backward(ctx.saved, ctx.weights, g_y.contiguous(), ctx.need_x)
        for t in keep:
            t.record_stream(side_stream())  # read by the helper stream after this frame is gone
        ctx.saved = None
        return (gx,) + tuple(gws)


# ---- the step split around the fused middle launch (hidvae_bottleneck_fwd) ------------------------------------------------------
# An activated tensor h = silu(pre) crosses the cut as the PAIR (pre, h): `pre` is the differentiable tensor, `h` a cache of its
# activation, so the consumer's input-gradient GEMM can apply silu'(pre) in its epilogue exactly as inside one MLPBodyFn and no
# extra elementwise launch appears at the cut.
class MLPFrontFn(Function):
    """Linear+SiLU stack whose LAST layer is activated too.  -> (pre_last [differentiable], silu(pre_last) [cache])"""

    @staticmethod
    def forward(ctx, x, *weights):
        ctx.set_materialize_grads(False)
        saved, h = [], x
        for w in weights:
            pre = torch.empty((h.shape[0], w.shape[0]), device=h.device, dtype=torch.float32)
            out = _C.gemm(_C.GEMM_NT, h, w, epilogue=_C.EPI_SILU, aux=pre)
            saved.append((h, pre))
            h = out
        pre_last = saved[-1][1]
        saved[-1] = (saved[-1][0], None)  # (an output: this node's backward does not need it, and holding it would be a cycle)
        ctx.saved, ctx.weights, ctx.need_x = saved, weights, ctx.needs_input_grad[0]
        ctx.mark_non_differentiable(h)
        return pre_last, h

    @staticmethod
    def backward(ctx, g_pre, _g_h):
        if g_pre is None:
            return (None,) * (1 + len(ctx.weights))
        gx, gws, _ = mlp_body_backward(ctx.saved, ctx.weights, g_pre.contiguous(), ctx.need_x)
        ctx.saved = None
        _C.phase_mark("bwd:encoder front done")
        return (gx,) + tuple(gws)


class MLPBackFn(Function):
    """MLPBodyFn on an input given as (pre_in [differentiable], act_in = silu(pre_in) [cache])."""

    @staticmethod
    def forward(ctx, pre_in, act_in, *weights):
        ctx.set_materialize_grads(False)
        y, saved = mlp_body_forward(act_in, weights, any(ctx.needs_input_grad))
        ctx.saved, ctx.weights, ctx.pre_in, ctx.need_x = saved, weights, pre_in, ctx.needs_input_grad[0]
        return y

    @staticmethod
    def backward(ctx, g_y):
        if g_y is None:
            return (None,) * (2 + len(ctx.weights))
        _C.phase_mark("bwd:decoder tail start")
        gx, gws, _ = mlp_body_backward(ctx.saved, ctx.weights, g_y.contiguous(), ctx.need_x, in_pre=ctx.pre_in)
        _C.phase_mark("bwd:decoder tail done")
        ctx.saved = None
        return (gx, None) + tuple(gws)


class BottleneckFn(Function):
    """encoder[-2:] + all L quantisation levels + decoder[:2] in one launch (csrc/rq.hip bottleneck_fwd_kernel); the backward is
    the same sequence of launches as the unfused path (paired Linear backward x4, rq_backward, codebook_grad).
    inputs : pre1 / h1 (the cut pair), W2, W3, Wd0, Wd1, config, prepared codebooks or None, `scratch` (a callable returning the
             caller-owned id-census table for this batch size, or None = no census in this launch), the L raw tables
    outputs: z, ids, emb_cat, emb_sum (non-differentiable here: its gradient is produced inside), qloss, pre_d1 / d1 (cut pair),
             embs_norm, p_unique (the debug statistics of h_rqvae.py:643-648 when the launch can carry the id census, else None)"""

    @staticmethod
    def forward(ctx, pre1, h1, W2, W3, Wd0, Wd1, normalize_input, mode, beta, normalize_flags, prepared, scratch, port, *tables):
        ctx.set_materialize_grads(False)
        ctx.port = port  # tagpath.HeadsGradPort or None: where emb_cat's gradient waits when the heads ran their own backward early
        if prepared is not None:
            cb, cc = prepared
            join_side()
        else:
            cb, cc = _C.codebook_prepare([t.detach() for t in tables], normalize_flags)
        census = scratch is not None and _C.census_eligible(cb.shape[0], cb.shape[1]) and os.environ.get("HIDVAE_FUSED_CENSUS", "1") != "0"
        o = _C.bottleneck_fwd(h1, W2.detach(), W3.detach(), cb, cc, normalize_input, mode, beta, Wd0.detach(), Wd1.detach(), id_stats=census,
                              scratch=scratch() if census else None)
        ctx.cfg = (normalize_input, mode, beta, tuple(normalize_flags))
        ctx.params = (W2, W3, Wd0, Wd1)
        ctx.tables = tables
        ctx.save_for_backward(pre1, h1, o["pre2"], o["h2"], o["y"], o["z"], o["ids"], o["emb_sum"], o["pre_d0"], o["d0"], cb, cc)
        # (ONE call: a second mark_non_differentiable would replace the first set, and emb_sum / d1 would silently keep an edge to this
        #  node -- which is what a split backward, HRqVae.dp_cut, trips over)
        ctx.mark_non_differentiable(*([o["ids"], o["emb_sum"], o["d1"]] + ([o["embs_norm"], o["p_unique"]] if census else [])))
        return o["z"], o["ids"], o["emb_cat"], o["emb_sum"], o["qloss"], o["pre_d1"], o["d1"], o.get("embs_norm"), o.get("p_unique")

    @staticmethod
    def backward(ctx, g_z, _g_ids, g_cat, _g_sum, g_q, g_pre_d1, _g_d1, _g_norm=None, _g_pu=None):
        normalize_input, mode, beta, flags = ctx.cfg
        W2, W3, Wd0, Wd1 = ctx.params
        pre1, h1, pre2, h2, y, z, ids, emb_sum, pre_d0, d0, cb, cc = ctx.saved_tensors
        g_sum = None
        gWd0 = gWd1 = None
        if g_pre_d1 is not None:
            dst, acc = grad_sink(Wd1)
            gWd1, g_pre_d0 = _C.linear_bwd(g_pre_d1.contiguous(), d0, Wd1, True, _C.EPI_DSILU, pre_d0, dW=dst, accumulate=acc)
            if dst is not None:
                gWd1 = None
            dst, acc = grad_sink(Wd0)
            gWd0, g_sum = _C.linear_bwd(g_pre_d0, emb_sum, Wd0, True, dW=dst, accumulate=acc)
            if dst is not None:
                gWd0 = None
        if g_cat is None and ctx.port is not None:
            g_cat = ctx.port.collect(as_slices=True)  # (the heads' prefix slices: summed inside the quantiser's backward launch)
        elif g_cat is not None:
            g_cat = g_cat.contiguous()
        g_z = g_z.contiguous() if g_z is not None else None
        g_y, dE = _C.rq_backward(y, z, cb, cc, normalize_input, mode, beta, ids, g_cat, g_sum, g_z, 1.0 if g_q is not None else 0.0, g_q)
        sinks = [getattr(t, "_hv_view", None) for t in ctx.tables]
        states = {bool(getattr(t, "_hv_written", False)) for t in ctx.tables}
        if all(torch.is_tensor(v) for v in sinks) and len(states) == 1:
            acc = [grad_sink(t)[1] for t in ctx.tables][0]
            _C.codebook_grad(ids, dE, [t.detach() for t in ctx.tables], cb, flags, grads=sinks, accumulate=acc)
            gE = [None] * len(sinks)
        else:
            gE = _C.codebook_grad(ids, dE, [t.detach() for t in ctx.tables], cb, flags)
        dst, acc = grad_sink(W3)
        gW3, g_pre2 = _C.linear_bwd(g_y, h2, W3, True, _C.EPI_DSILU, pre2, dW=dst, accumulate=acc)
        if dst is not None:
            gW3 = None
        dst, acc = grad_sink(W2)
        gW2, g_pre1 = _C.linear_bwd(g_pre2, h1, W2, ctx.needs_input_grad[0], _C.EPI_DSILU, pre1, dW=dst, accumulate=acc)
        if dst is not None:
            gW2 = None
        return (g_pre1, None, gW2, gW3, gWd0, gWd1, None, None, None, None, None, None, None) + tuple(gE)


class L2NormFn(Function):
    """F.normalize(x, dim=-1, eps) (modules/normalize.py:7-8)."""

    @staticmethod
    def forward(ctx, x, eps):
        ctx.set_materialize_grads(False)
        out, norms = _C.l2norm_fwd(x, eps)
        ctx.save_for_backward(out, norms)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        out, norms = ctx.saved_tensors
        return _C.l2norm_bwd(g.contiguous(), out, norms, ctx.eps), None


class RQFn(Function):
    """All L quantisation levels in one kernel (h_rqvae.py:515-552 + quantize.py:100-153 of the reference).

    inputs : y [B,32] (pre-normalisation iff normalize_input), the L raw codebook tables
    outputs: z, ids (int64, non-differentiable), emb_cat [B,L*32], emb_sum [B,32], qloss [B], res_cat"""

    @staticmethod
    def forward(ctx, y, normalize_input, mode, training, beta, normalize_flags, want_res, prepared, port, *tables):
        ctx.set_materialize_grads(False)
        ctx.port = port  # tagpath.HeadsGradPort or None (see BottleneckFn)
        distance = _C.DIST_L2
        if isinstance(mode, tuple):  # (forward mode, distance): Quantize on its own with QuantizeDistance.COSINE (quantize.py:115-119)
            mode, distance = mode
        if prepared is not None:  # effective codebooks were computed on the helper stream beside the encoder
            cb, cc = prepared
            join_side()
        else:
            cb, cc = _C.codebook_prepare([t.detach() for t in tables], normalize_flags)
        z, ids, emb_cat, emb_sum, res, qloss = _C.rq_forward(y, cb, cc, normalize_input, mode, training, beta, want_res=want_res,
                                                             distance=distance)
        ctx.cfg = (normalize_input, mode, training, beta, tuple(normalize_flags))
        ctx.tables = tables
        # (save_for_backward, not attributes: z and ids are OUTPUTS of this node, and an output held by its own ctx is a reference
        #  cycle -- a forward that is never followed by a backward, e.g. the k-means warm-up pass, would then keep its whole autograd
        #  graph, AccumulateGrad nodes included, alive until the garbage collector runs; those stale nodes sit on the stream of that
        #  forward and break a later HIP-graph capture)
        ctx.save_for_backward(y, z, ids, cb, cc)
        if res is None:
            res = torch.empty(0, device=y.device)
        ctx.mark_non_differentiable(ids, res)  # (one call: a second one would replace the first set)
        return z, ids, emb_cat, emb_sum, qloss, res

    @staticmethod
    def backward(ctx, g_z, _g_ids, g_cat, g_sum, g_q, _g_res):
        normalize_input, mode, training, beta, flags = ctx.cfg
        if not training:
            raise RuntimeError("the eval branch of Quantize (o = codebook[ids]) is not differentiated on the fused path")
        y, z, ids, cb, cc = ctx.saved_tensors
        if g_cat is None and ctx.port is not None:
            g_cat = ctx.port.collect()
        if g_cat is not None:
            g_cat = g_cat.contiguous()
        if g_sum is not None:
            g_sum = g_sum.contiguous()
        if g_z is not None:
            g_z = g_z.contiguous()
        g_y, dE = _C.rq_backward(y, z, cb, cc, normalize_input, mode, beta, ids, g_cat, g_sum, g_z,
                                 1.0 if g_q is not None else 0.0, g_q)
        main, side = torch.cuda.current_stream(), side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):  # the per-code gather runs beside the encoder's backward
            sinks = [getattr(t, "_hv_view", None) for t in ctx.tables]
            states = {bool(getattr(t, "_hv_written", False)) for t in ctx.tables}
            if all(torch.is_tensor(v) for v in sinks) and len(states) == 1:  # every table owns a flat-gradient slot
                acc = [grad_sink(t)[1] for t in ctx.tables][0]
                _C.codebook_grad(ids, dE, [t.detach() for t in ctx.tables], cb, flags, grads=sinks, accumulate=acc)
                gE = [None] * len(sinks)
            else:
                gE = _C.codebook_grad(ids, dE, [t.detach() for t in ctx.tables], cb, flags)
        for t in (dE, ids, cb):
            t.record_stream(side)
        for t in gE:
            if t is not None:
                t.record_stream(main)
        _join_after_backward()
        return (g_y, None, None, None, None, None, None, None, None) + tuple(gE)


class ReconFn(Function):
    """decoder tail: recon[b] = sum_j (normalize(y)[b,j] - x[b,j])^2  (encoder.py:32 + loss.py:11-12); with n_cat > 0 the last
    n_cat columns of the row-normalised y enter as BCE-with-logits and the head is normalised once more (h_rqvae.py:610-613)."""

    @staticmethod
    def forward(ctx, y, x, n_cat=0):
        ctx.set_materialize_grads(False)
        recon, _, _ = _C.recon_fwd_bwd(y, x, n_cat=n_cat)
        ctx.save_for_backward(y, x)
        ctx.n_cat = n_cat
        return recon

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        y, x = ctx.saved_tensors
        _, _, g_y = _C.recon_fwd_bwd(y, x, gscale=1.0, gscale_items=g, want_grad=True, n_cat=ctx.n_cat)
        return g_y, None, None


class CatReconRowsFn(Function):
    """CategoricalReconstructionLoss.forward on a given x_hat (reference loss.py:15-33): one launch each way."""

    @staticmethod
    def forward(ctx, x_hat, x, n_cat):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x_hat, x)
        ctx.n_cat = n_cat
        return _C.cat_recon_rows(x_hat, x, n_cat)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None
        x_hat, x = ctx.saved_tensors
        return _C.cat_recon_rows(x_hat, x, ctx.n_cat, g=g.contiguous()), None, None


class TotalLossFn(Function):
    """loss = mean(recon) + mean(qloss) + w_a*align + w_p*pred + w_u*uniq  (h_rqvae.py:634-640), with the uniqueness
    term evaluated in the same launch exactly as the reference calls it ([L,B] ids) and the per-level tag
    losses averaged the reference's way (sum over levels / n_layers, h_rqvae.py:561-563).
    Inputs after the hyper-parameters: n_tag alignment scalars, n_tag prediction scalars, n_tag accuracy scalars.
    Returns (loss, uniq, tagstats); only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, recon, qloss, z, ids, uniq_weight, uniq_margin, w_a, w_p, w_u, n_tag, tag_div, *tag_scalars):
        ctx.set_materialize_grads(False)
        want = z is not None and ctx.needs_input_grad[2]
        aligns, preds, accs = list(tag_scalars[:n_tag]), list(tag_scalars[n_tag:2 * n_tag]), list(tag_scalars[2 * n_tag:3 * n_tag])
        loss, uniq, g_rows, tagstats = _C.total_loss(recon, qloss, [t.detach() for t in aligns], [t.detach() for t in preds],
                                                     [t.detach() for t in accs], tag_div, ids, z, uniq_weight, uniq_margin, w_a, w_p,
                                                     w_u, want)
        ctx.meta = (recon.shape[0], ids.shape[1] if ids is not None else 0, w_a, w_p, w_u, n_tag, tag_div, z is not None)
        ctx.g_rows = g_rows
        if tagstats is None:
            tagstats = torch.empty(0, device=loss.device)
        ctx.mark_non_differentiable(uniq, tagstats)
        return loss, uniq, tagstats

    @staticmethod
    def backward(ctx, g, _g_uniq, _g_stats):
        B, L, w_a, w_p, w_u, n_tag, tag_div, has_z = ctx.meta
        if g is None:
            return (None,) * (11 + 3 * n_tag)
        scal, g_z = _C.total_loss_bwd(g.contiguous(), B, L, w_a / tag_div, w_p / tag_div, w_u, ctx.g_rows,
                                      want_gz=has_z and ctx.g_rows is not None)
        per_item = scal[0].expand(B)  # stride-0 view: the kernels downstream read one device scalar
        tags = (scal[1],) * n_tag + (scal[2],) * n_tag + (None,) * n_tag
        return (per_item, per_item, g_z, None, None, None, None, None, None, None, None) + tags


class StepLossFn(Function):
    """ReconFn + TotalLossFn as HRqVae.forward pairs them, one launch each way (hidvae_loss_fwd / hidvae_loss_bwd).
    Inputs: decoder body output y, the batch x, then as TotalLossFn.  Returns (loss, recon, uniq, tagstats, summary); only `loss`
    is differentiable (the reconstruction term's gradient travels inside it)."""

    @staticmethod
    def forward(ctx, y, x, qloss, z, ids, uniq_weight, uniq_margin, w_a, w_p, w_u, n_tag, tag_div, n_cat, *tag_scalars):
        ctx.set_materialize_grads(False)
        ctx.expect_g = 0.0
        if isinstance(n_cat, tuple):  # (n_cat, the loss gradient the tag heads' early backward was seeded with: checked in the backward launch)
            n_cat, ctx.expect_g = n_cat
        ctx.n_cat = n_cat
        want = z is not None and ctx.needs_input_grad[3]
        aligns, preds, accs = list(tag_scalars[:n_tag]), list(tag_scalars[n_tag:2 * n_tag]), list(tag_scalars[2 * n_tag:3 * n_tag])
        loss, recon, uniq, g_rows, tagstats, summary = _C.loss_fwd(y, x, qloss.detach(), [t.detach() for t in aligns], [t.detach() for t in preds],
                                                          [t.detach() for t in accs], tag_div, ids, z, uniq_weight, uniq_margin, w_a,
                                                          w_p, w_u, want, n_cat=n_cat)
        ctx.meta = (y.shape[0], ids.shape[1] if ids is not None else 0, w_a, w_p, w_u, n_tag, tag_div, z is not None)
        ctx.g_rows = g_rows
        ctx.save_for_backward(y, x)
        if tagstats is None:
            tagstats = torch.empty(0, device=loss.device)
        ctx.mark_non_differentiable(recon, uniq, tagstats, summary)
        return loss, recon, uniq, tagstats, summary

    @staticmethod
    def backward(ctx, g, _g_recon, _g_uniq, _g_stats, _g_summary):
        B, L, w_a, w_p, w_u, n_tag, tag_div, has_z = ctx.meta
        if g is None:
            return (None,) * (13 + 3 * n_tag)
        y, x = ctx.saved_tensors
        g_y, scal, g_z = _C.loss_bwd(g.contiguous(), y, x, L, w_a / tag_div, w_p / tag_div, w_u, ctx.g_rows,
                                     want_gz=has_z and ctx.g_rows is not None, n_cat=ctx.n_cat, expect_g=ctx.expect_g)
        per_item = scal[0].expand(B)  # stride-0 view: the kernels downstream read one device scalar
        tags = (scal[1],) * n_tag + (scal[2],) * n_tag + (None,) * n_tag
        return (g_y, None, per_item, g_z, None, None, None, None, None, None, None, None, None) + tags


class UniqLossFn(Function):
    """SemanticIdUniquenessLoss.forward(sem_ids [n,m], encoded_features) with the reference's literal semantics (h_rqvae.py:41-105):
    rows of the id matrix that agree everywhere are pushed apart.  Differentiable w.r.t. encoded_features: the gradient lands on
    rows [0, n) -- the same rule hidvae_total_loss_bwd applies inside the fused step."""

    @staticmethod
    def forward(ctx, ids_t, z, weight, margin):
        ctx.set_materialize_grads(False)
        want = ctx.needs_input_grad[1]
        m, n = ids_t.shape  # n id vectors of length m
        if n > _C.MAX_LEVELS:
            raise NotImplementedError(f"SemanticIdUniquenessLoss on the HIP path compares at most {_C.MAX_LEVELS} id vectors (got {n}): "
                                      "HRqVae.forward calls it with one vector per LEVEL (the [L,B] ids of h_rqvae.py:630-631)")
        if z.shape[0] < n:
            raise IndexError(f"encoded_features has {z.shape[0]} rows, the id matrix {n}")  # the reference's encoded_features[b] would raise
        if m < n:  # the kernel takes the id-vector length as the number of feature rows available: repeat the vectors' entries
            ids_t = ids_t.repeat((n + m - 1) // m, 1).contiguous()  # (equality of whole vectors is unchanged)
        loss, g_rows = _C.uniq_loss(ids_t, z.detach().contiguous(), weight, margin, want_grad=want)
        ctx.shape = tuple(z.shape)
        if want:
            ctx.save_for_backward(g_rows)
        return loss

    @staticmethod
    def backward(ctx, g):
        if g is None or not ctx.saved_tensors:
            return None, None, None, None
        (g_rows,) = ctx.saved_tensors
        B, L = ctx.shape[0], g_rows.shape[0]
        _, g_z = _C.total_loss_bwd(g.contiguous(), B, L, 0.0, 0.0, 1.0, g_rows, want_gz=True)  # g * g_rows on rows < L, 0 elsewhere
        return None, g_z, None, None


class SqDiffRowsFn(Function):
    """out[m] = sum_j (a-b)^2 with independently scaled gradients for the two operands: ReconstructionLoss (scales 1, 1) and the
    stop-gradient halves of QuantizeLoss (reference loss.py:7-12, 36-44)."""

    @staticmethod
    def forward(ctx, a, b, scale_a, scale_b, extra):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(a, b)
        ctx.scales = (scale_a, scale_b)
        return _C.sqdiff_rows(a, b, extra)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None, None
        a, b = ctx.saved_tensors
        ga, gb = _C.sqdiff_rows_bwd(g.contiguous(), a, b, ctx.scales[0], ctx.scales[1], ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return ga, gb, None, None, None


class LinearFn(Function):
    """y = act(x W^T + b) [* keep_mask * keep_scale]  -- nn.Linear with the activation (and dropout) fused into the GEMM
    epilogue.  act is an _C.EPI_* forward code.  Backward: g_pre = g * act'(.) through the elementwise kernel, then the
    input gradient (NN GEMM) and weight gradient (TN GEMM, split-K) in one paired launch, bias gradient (column sums).
    Two statically wired fusions for chains (the caller, who composes the chain, sets them -- both need the tensor in question to
    have exactly ONE consumer):
      act_bwd_done  the incoming gradient was already taken through THIS layer's activation / dropout by whoever produced it (the
                    next layer's input-gradient epilogue, or the LayerNorm backward that follows): no elementwise launch here;
      dx_gate       scale s: the INPUT x of this layer is relu(.) * keep * s (the previous Linear -> ReLU -> Dropout), and the input
                    gradient leaves this launch already through that gate, (g W) * (x > 0 ? s : 0)."""

    @staticmethod
    def forward(ctx, x, w, b, act, keep_mask=None, keep_scale=1.0, act_bwd_done=False, dx_gate=None, computed=None):
        """computed: this layer's output, already produced by a fused launch (tagpath: hidvae_predictor_fwd) -- nothing is launched here,
        the node only takes its place on the tape"""
        ctx.set_materialize_grads(False)
        need = any(ctx.needs_input_grad)
        pre = None
        if computed is not None:
            if act in (_C.EPI_SILU, _C.EPI_GELU) or tuple(computed.shape) != (x.shape[0], w.shape[0]):
                raise RuntimeError("LinearFn: a precomputed output needs a ReLU / identity layer of the same shape")
            y = computed
        else:
            if need and act in (_C.EPI_SILU, _C.EPI_GELU):
                pre = torch.empty((x.shape[0], w.shape[0]), device=x.device, dtype=torch.float32)
            y = _C.gemm(_C.GEMM_NT, x, w, bias=b, epilogue=act, aux=pre, mask=keep_mask, mask_scale=keep_scale, split_k=0)
        ctx.act, ctx.keep_scale = act, keep_scale
        ctx.has_bias = b is not None
        ctx.w_param, ctx.b_param = w, b  # (the objects themselves: a flat-gradient slot hangs off the Parameter)
        ctx.need_x = ctx.needs_input_grad[0]
        ctx.act_bwd_done, ctx.dx_gate = bool(act_bwd_done), dx_gate
        if keep_mask is not None and not torch.is_tensor(keep_mask):  # a _C.DropSpec: decided inside the launch, never stored
            if act != _C.EPI_RELU:
                raise NotImplementedError("in-kernel dropout is wired for Linear -> ReLU -> Dropout (the gate is read off the output)")
            ctx.drop_after_relu = True
            keep_mask = None
        else:
            ctx.drop_after_relu = False
        ctx.save_for_backward(x, w, pre if pre is not None else y, keep_mask)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 9
        x, w, ref, keep_mask = ctx.saved_tensors
        g = g.contiguous()
        if not ctx.act_bwd_done and (ctx.act != _C.EPI_NONE or keep_mask is not None):
            # (ReLU: the kernel gates on ref = y > 0 and only needs a non-NULL mask argument to apply keep_scale)
            g = _C.act_bwd(g, ref, ctx.act, ref if ctx.drop_after_relu else keep_mask, ctx.keep_scale)
        epi, aux, dxs = (_C.EPI_DRELU, x, float(ctx.dx_gate)) if (ctx.dx_gate is not None and ctx.need_x) else (_C.EPI_NONE, None, 1.0)
        dst, acc = grad_sink(ctx.w_param)
        if ctx.has_bias:  # weight, input and bias gradients share one launch
            bdst, bacc = grad_sink(ctx.b_param)
            gw, gx, gb = _C.linear_bwd(g, x, w, ctx.need_x, epi, aux, dW=dst, accumulate=acc, bias=True, db=bdst, accumulate_db=bacc, dx_scale=dxs)
            if bdst is not None:
                gb = None
        else:
            gw, gx = _C.linear_bwd(g, x, w, ctx.need_x, epi, aux, dW=dst, accumulate=acc, dx_scale=dxs)
            gb = None
        if dst is not None:
            gw = None
        return gx, gw, gb, None, None, None, None, None, None
