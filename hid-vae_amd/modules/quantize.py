"""One residual-quantisation level (reference modules/quantize.py:17-153) on the fused HIP RQ kernel."""
from enum import Enum
from typing import NamedTuple

import torch
from torch import Tensor, nn

from .. import _C, gin_compat as gin
from ..ops import RQFn
from .loss import QuantizeLoss
from .normalize import L2NormalizationLayer


@gin.constants_from_enum
class QuantizeForwardMode(Enum):
    GUMBEL_SOFTMAX = 1
    STE = 2
    ROTATION_TRICK = 3


class QuantizeDistance(Enum):
    L2 = 1
    COSINE = 2


class QuantizeOutput(NamedTuple):
    embeddings: Tensor
    ids: Tensor
    loss: Tensor


class Quantize(nn.Module):
    def __init__(self, embed_dim: int, n_embed: int, do_kmeans_init: bool = True, codebook_normalize: bool = False,
                 sim_vq: bool = False, commitment_weight: float = 0.25,
                 forward_mode: QuantizeForwardMode = QuantizeForwardMode.GUMBEL_SOFTMAX,
                 distance_mode: QuantizeDistance = QuantizeDistance.L2):
        super().__init__()
        if distance_mode not in (QuantizeDistance.L2, QuantizeDistance.COSINE):
            raise Exception("Unsupported Quantize distance mode.")  # quantize.py:121
        self.embed_dim, self.n_embed = embed_dim, n_embed
        self.embedding = nn.Embedding(n_embed, embed_dim)
        self.forward_mode, self.distance_mode = forward_mode, distance_mode
        self.do_kmeans_init, self.kmeans_initted = do_kmeans_init, False
        self.codebook_normalize, self.sim_vq = codebook_normalize, sim_vq
        self.out_proj = nn.Sequential(
            nn.Linear(embed_dim, embed_dim, bias=False) if sim_vq else nn.Identity(),
            L2NormalizationLayer(dim=-1) if codebook_normalize else nn.Identity())
        self.quantize_loss = QuantizeLoss(commitment_weight)
        nn.init.uniform_(self.embedding.weight)  # reference quantize.py:86-89

    @property
    def weight(self) -> Tensor:
        return self.embedding.weight

    @property
    def device(self):
        return self.embedding.weight.device

    @torch.no_grad()
    def _kmeans_init(self, x) -> None:
        from ..init.kmeans import kmeans_init_
        kmeans_init_(self.embedding.weight, x=x)
        self.kmeans_initted = True

    def table(self):
        """Codebook table handed to the fused kernel: raw embedding, or its sim-VQ projection (quantize.py:70-73).
        Row normalisation (quantize.py:72) is applied inside the kernel's codebook-prepare step."""
        if self.sim_vq:
            from ..ops import LinearFn
            return LinearFn.apply(self.embedding.weight, self.out_proj[0].weight, None, _C.EPI_NONE)
        return self.embedding.weight

    def get_item_embeddings(self, item_ids) -> Tensor:
        cb, _ = _C.codebook_prepare([self.table().detach().contiguous()], [self.codebook_normalize])
        return cb[0][item_ids]

    def forward(self, x, temperature) -> QuantizeOutput:
        assert x.shape[-1] == self.embed_dim
        if self.do_kmeans_init and not self.kmeans_initted:
            self._kmeans_init(x=x)
        mode = self.forward_mode.value
        if self.training and mode == QuantizeForwardMode.GUMBEL_SOFTMAX.value:
            from ..gumbel_path import gumbel_level
            return QuantizeOutput(*gumbel_level(self, x, temperature, rand=getattr(self, "rand", None),
                                                cosine=self.distance_mode == QuantizeDistance.COSINE))
        if self.distance_mode == QuantizeDistance.COSINE:  # -(x/|x| . c)/|c| ranks the codes (quantize.py:115-119); all else as L2
            mode = (mode, _C.DIST_COSINE)
        _, ids, emb_cat, _, qloss, _ = RQFn.apply(x.contiguous(), False, mode, self.training, self.quantize_loss.commitment_weight,
                                                  (self.codebook_normalize,), False, None, None, self.table())
        return QuantizeOutput(embeddings=emb_cat, ids=ids[:, 0], loss=qloss)
