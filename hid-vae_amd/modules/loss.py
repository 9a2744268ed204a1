"""Loss modules of the tokenizer (reference modules/loss.py).  Each forward is a C-ABI launch; the classes keep
the reference's constructor arguments and the attributes train_hidvae.py pokes (loss.py:96-102 knobs)."""
from torch import nn

from ..ops import CatReconRowsFn, ReconFn, SqDiffRowsFn


def _rows(t):
    """[..., N] -> a 2-D view [M, N] with a contiguous last dim (the row kernels take a row stride)"""
    t = t.float()
    t = t.reshape(-1, t.shape[-1])
    return t if t.stride(-1) == 1 else t.contiguous()


class ReconstructionLoss(nn.Module):
    """sum_j (x_hat - x)^2 per item (reference loss.py:7-12): one launch forward, one backward (hidvae_sqdiff_rows[_bwd]).
    HRqVae.forward itself uses `fused`, which takes the decoder output BEFORE its trailing L2 normalisation and folds the
    normalisation into the same kernel."""

    def forward(self, x_hat, x):
        out = SqDiffRowsFn.apply(_rows(x_hat), _rows(x), 1.0, 1.0, 0.0)
        return out.reshape(x_hat.shape[:-1])

    @staticmethod
    def fused(decoder_body_out, x):
        return ReconFn.apply(decoder_body_out.contiguous(), x.contiguous())


class CategoricalReconstructionLoss(nn.Module):
    """Squared error on all but the last n_cat_feats columns + binary cross-entropy with logits on those (reference loss.py:15-33):
    one launch each way (hidvae_cat_recon_rows).  `fused` takes the decoder output before its trailing L2 normalisation and folds
    both normalisations of the forward (encoder.py:32, h_rqvae.py:610) into the same kernel."""

    def __init__(self, n_cat_feats: int) -> None:
        super().__init__()
        self.reconstruction_loss = ReconstructionLoss()
        self.n_cat_feats = n_cat_feats

    def forward(self, x_hat, x):
        out = CatReconRowsFn.apply(_rows(x_hat), _rows(x), int(self.n_cat_feats))
        return out.reshape(x_hat.shape[:-1])

    def fused(self, decoder_body_out, x):
        return ReconFn.apply(decoder_body_out.contiguous(), x.contiguous(), int(self.n_cat_feats))


CategoricalReconstuctionLoss = CategoricalReconstructionLoss  # the name h_rqvae.py:14 / rqvae.py import (a typo in the reference)


class QuantizeLoss(nn.Module):
    """|sg(q) - v|^2 + beta |q - sg(v)|^2 (reference loss.py:36-44).  Inside HRqVae the fused RQ kernel evaluates it; called on its
    own it is one launch each way: both terms have the value s = sum_j (q-v)^2, the stop-gradients only route the gradient
    (query gets beta * 2(q-v), value gets 2(v-q))."""

    def __init__(self, commitment_weight: float = 1.0):
        super().__init__()
        self.commitment_weight = commitment_weight

    def forward(self, query, value):
        cw = float(self.commitment_weight)
        return SqDiffRowsFn.apply(_rows(query), _rows(value), cw, 1.0, cw).reshape(query.shape[:-1])


class TagAlignmentLoss(nn.Module):
    """InfoNCE between concatenated codebook embeddings and projected tag embeddings (reference loss.py:48-85)."""

    def __init__(self, alignment_weight: float = 1.0, temperature: float = 0.1):
        super().__init__()
        self.alignment_weight = alignment_weight
        self.temperature = temperature

    def forward(self, codebook_emb, tag_emb, layer_idx: int):
        from ..tagpath import InfoNCEFn
        scale = self.alignment_weight * (1.0 / (layer_idx * 0.5 + 1))
        return InfoNCEFn.apply(codebook_emb, tag_emb.contiguous(), self.temperature, scale)


class TagPredictionLoss(nn.Module):
    """Focal / CE tag classification loss with mixup and label smoothing (reference loss.py:89-265).
    As in the reference the model always calls it with layer_idx = 0 and class_counts stays None."""

    def __init__(self, use_focal_loss: bool = False, focal_params: dict = None, class_counts: dict = None):
        super().__init__()
        self.use_focal_loss = use_focal_loss
        self.focal_params = focal_params or {"gamma": 2.0, "alpha": 0.25}
        self.class_counts = class_counts
        self.use_label_smoothing = True
        self.label_smoothing_alpha = 0.1
        self.use_mixup = True
        self.mixup_alpha = 0.2
        self.weight_scheduler = None
        self.rand = None  # injectable randomness provider (hidvae_amd.rand); None = device RNG

    def forward(self, pred_logits, target_indices, layer_idx: int = 0):
        from ..tagpath import tag_prediction_loss
        if self.class_counts is not None:
            raise NotImplementedError("class-frequency weighted focal loss is unreachable in the reference (SURVEY Q5)")
        return tag_prediction_loss(self, pred_logits, target_indices, layer_idx)
