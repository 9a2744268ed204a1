"""GUMBEL_SOFTMAX training branch of Quantize (reference modules/quantize.py:125-130 + distributions/gumbel.py:8-18).

No shipped config selects it (both gin files bind ROTATION_TRICK) although it is the HRqVae constructor default, so it is
composed level by level from the MFMA GEMM entry points and the row kernels of csrc/gumbel.hip instead of living in the
fused RQ kernel.  Every codebook row receives gradient here (the soft assignment touches all K codes).  Any embed_dim the
quantiser takes (<= 64); the cosine ranking of a stand-alone Quantize too (GumbelCosineLevelFn)."""
import torch
from torch.autograd import Function

from . import _C
from .ops import L2NormFn
from .tagpath import AddFn


class GumbelLevelFn(Function):
    """(x [B,D], cb [K,D] effective codebook, U [B,K] uniform draws) -> (emb [B,D], ids [B], loss [B]); any D <= 64."""

    @staticmethod
    def forward(ctx, x, cb, U, temperature, beta):
        ctx.set_materialize_grads(False)
        x, cb = x.contiguous(), cb.contiguous()
        S = _C.gemm(_C.GEMM_NT, x, cb, split_k=0)           # x cb^T
        cc = _C.codebook_prepare([cb], [False])[1][0]       # |c_k|^2
        ids = _C.gumbel_rows_fwd(S, x, cc, U.contiguous(), temperature)  # S -> P
        emb = _C.gemm(_C.GEMM_NN, S, cb, split_k=0)          # P cb
        loss = _C.gumbel_loss(x, emb, beta)
        ctx.save_for_backward(x, cb, S, emb)
        ctx.cfg = (temperature, beta)
        ctx.mark_non_differentiable(ids)
        return emb, ids, loss

    @staticmethod
    def backward(ctx, g_out, _g_ids, g_l):
        x, cb, P, emb = ctx.saved_tensors
        temperature, beta = ctx.cfg
        if g_out is None and g_l is None:
            return None, None, None, None, None
        g_emb = _C.gumbel_gemb(g_out.contiguous() if g_out is not None else None, g_l, x, emb)
        gP = _C.gemm(_C.GEMM_NT, g_emb, cb, split_k=0)        # [B,K]
        g_cb = _C.gemm(_C.GEMM_TN, P, g_emb, split_k=0)       # P^T g_emb
        g_xx = _C.gumbel_rows_bwd(P, gP, temperature)         # gP -> g_S
        g_x = _C.gemm(_C.GEMM_NN, gP, cb, split_k=0)          # g_S cb
        _C.gemm(_C.GEMM_TN, gP, x, out=g_cb, split_k=0, accumulate=True)  # + g_S^T x
        _C.gumbel_finish(g_x, x, emb, g_xx, g_l, beta, g_cb, cb, _C.colsum(gP))
        return g_x, g_cb, None, None, None


class GumbelCosineLevelFn(Function):
    """QuantizeDistance.COSINE under GUMBEL_SOFTMAX (reference quantize.py:115-119,125-130): dist = -(x^ . c^) of the normalised query
    and code rows, weights = softmax((-dist + G)/T), emb = weights @ codebook (the codebook itself, not its normalised rows).
    (xh [B,D] = x / |x|, ch [K,D] = cb rows / |cb row| -- both differentiable outside --, cb [K,D], x [B,D], U) -> (emb, ids, loss)."""

    @staticmethod
    def forward(ctx, xh, ch, cb, x, U, temperature, beta):
        ctx.set_materialize_grads(False)
        xh, ch, cb, x = xh.contiguous(), ch.contiguous(), cb.contiguous(), x.contiguous()
        S = _C.gemm(_C.GEMM_NT, xh, ch, split_k=0)                                  # x^ c^T
        ids = _C.gumbel_rows_fwd(S, x, None, U.contiguous(), temperature, cosine=True)  # S -> P
        emb = _C.gemm(_C.GEMM_NN, S, cb, split_k=0)                                 # P cb
        loss = _C.gumbel_loss(x, emb, beta)
        ctx.save_for_backward(xh, ch, cb, x, S, emb)
        ctx.cfg = (temperature, beta)
        ctx.mark_non_differentiable(ids)
        return emb, ids, loss

    @staticmethod
    def backward(ctx, g_out, _g_ids, g_l):
        xh, ch, cb, x, P, emb = ctx.saved_tensors
        temperature, beta = ctx.cfg
        if g_out is None and g_l is None:
            return (None,) * 7
        g_emb = _C.gumbel_gemb(g_out.contiguous() if g_out is not None else None, g_l, x, emb)
        gP = _C.gemm(_C.GEMM_NT, g_emb, cb, split_k=0)        # [B,K]
        g_cb = _C.gemm(_C.GEMM_TN, P, g_emb, split_k=0)       # P^T g_emb
        _C.gumbel_rows_bwd(P, gP, temperature)                # gP -> 2/T P (gP - sum P gP): the L2 form's g_S; the cosine form's is half
        gS = _C.mul(gP, torch.full_like(gP, 0.5))
        g_xh = _C.gemm(_C.GEMM_NN, gS, ch, split_k=0)
        g_ch = _C.gemm(_C.GEMM_TN, gS, xh, split_k=0)
        # the commitment term's gradient to x: g_l 2 beta (x - emb)   (gumbel_finish with zero |x|^2 / |c|^2 parts)
        g_x = torch.zeros_like(x)
        zero_b, zero_k = torch.zeros(x.shape[0], device=x.device), torch.zeros(cb.shape[0], device=x.device)
        _C.gumbel_finish(g_x, x, emb, zero_b, g_l, beta, g_cb, cb, zero_k)
        return g_xh, g_ch, g_cb, g_x, None, None, None


class SubFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return _C.binary(2, a, b)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        g = g.contiguous()
        return g, _C.binary(2, torch.zeros_like(g), g)


def _effective(layer):
    cb = layer.table()
    if layer.codebook_normalize:
        cb = L2NormFn.apply(cb.contiguous(), 1e-12)
    return cb


def gumbel_level(layer, x, temperature, rand=None, cosine=False):
    from .rand import default_rand
    rand = rand or default_rand()
    U = rand.gumbel_u((x.shape[0], layer.n_embed), x.device)
    cb = _effective(layer)
    if cosine:  # (the reference divides by the plain norms, quantize.py:116-118; a zero row is as undefined there as here)
        return GumbelCosineLevelFn.apply(L2NormFn.apply(x.contiguous(), 1e-12), L2NormFn.apply(cb.contiguous(), 1e-12), cb, x, U,
                                         float(temperature), layer.quantize_loss.commitment_weight)
    return GumbelLevelFn.apply(x, cb, U, float(temperature), layer.quantize_loss.commitment_weight)


def gumbel_all_levels(model, y, normalize_input):
    """-> z, ids [B,L], emb_cat [B,L*D], emb_sum [B,D], qloss [B], res_cat [B,L*D]   (same tuple as ops.RQFn)"""
    rand = model._rand()
    z = L2NormFn.apply(y.contiguous(), 1e-12) if normalize_input else y
    res, embs, ids, ress = z, [], [], []
    qloss = esum = None
    t = getattr(model, "_gumbel_t", 1.0)
    for layer in model.layers:
        ress.append(res)
        emb, idl, loss = gumbel_level(layer, res, t, rand)
        embs.append(emb)
        ids.append(idl)
        qloss = loss if qloss is None else AddFn.apply(qloss.unsqueeze(1), loss.unsqueeze(1)).squeeze(1)
        esum = emb if esum is None else AddFn.apply(esum, emb)
        res = SubFn.apply(res, emb)
    # (assembling the side-by-side buffers is device plumbing; this branch is not on any config's path)
    return z, torch.stack(ids, dim=1), torch.cat(embs, dim=1), esum, qloss, torch.cat([r.detach() for r in ress], dim=1)
