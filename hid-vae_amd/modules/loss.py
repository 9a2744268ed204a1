"""Loss modules of the tokenizer (reference modules/loss.py).  Each forward is a C-ABI launch; the classes keep
the reference's constructor arguments and the attributes train_hidvae.py pokes (loss.py:96-102 knobs)."""
from torch import nn

from ..ops import ReconFn


class ReconstructionLoss(nn.Module):
    """sum_j (x_hat - x)^2 per item (reference loss.py:7-12).  `x_hat_pre` is the decoder output BEFORE its trailing
    L2 normalisation when called through `fused`, which folds the normalisation into the same kernel."""

    def forward(self, x_hat, x):
        raise NotImplementedError("ReconstructionLoss on an already-normalised x_hat is not on the hot path; "
                                  "HRqVae.forward uses ReconstructionLoss.fused(decoder_body_out, x)")

    @staticmethod
    def fused(decoder_body_out, x):
        return ReconFn.apply(decoder_body_out.contiguous(), x.contiguous())


class QuantizeLoss(nn.Module):
    """|sg(q) - v|^2 + beta |q - sg(v)|^2 (reference loss.py:36-44); evaluated inside the fused RQ kernel."""

    def __init__(self, commitment_weight: float = 1.0):
        super().__init__()
        self.commitment_weight = commitment_weight


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
