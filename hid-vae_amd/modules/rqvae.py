"""Plain RqVae (reference modules/rqvae.py:37-164): the tokenizer without the tag heads and without the uniqueness term --
a strict subset of HRqVae's step, so it runs on the same launches (MLP GEMMs, fused L-level quantisation or the fused middle
launch, reconstruction + total loss, id statistics).  Same constructor kwargs, methods and state-dict keys as the reference
(8 MLP weights + L codebooks [+ L sim-VQ projections])."""
from collections import namedtuple
from typing import List

import torch
from torch import Tensor, nn

from .. import _C
from .encoder import MLP
from .h_rqvae import HRqVae, SemanticIdUniquenessLoss
from .loss import CategoricalReconstructionLoss, ReconstructionLoss
from .quantize import Quantize, QuantizeForwardMode

RqVaeOutput = namedtuple("RqVaeOutput", ("embeddings", "residuals", "sem_ids", "quantize_loss"))
RqVaeComputedLosses = namedtuple("RqVaeComputedLosses", ("loss", "reconstruction_loss", "rqvae_loss", "embs_norm", "p_unique_ids"))


class RqVae(HRqVae):
    def __init__(self, input_dim: int, embed_dim: int, hidden_dims: List[int], codebook_size: int, codebook_kmeans_init: bool = True,
                 codebook_normalize: bool = False, codebook_sim_vq: bool = False,
                 codebook_mode: QuantizeForwardMode = QuantizeForwardMode.GUMBEL_SOFTMAX, n_layers: int = 3,
                 commitment_weight: float = 0.25, n_cat_features: int = 18) -> None:
        self._config = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        nn.Module.__init__(self)
        if not 0 <= n_cat_features < input_dim:
            raise ValueError(f"n_cat_features={n_cat_features} must leave at least one of the {input_dim} input columns non-categorical")
        _C.check_embed_dim(embed_dim)  # 32: the fused kernels; other multiples of 4 up to 64 (rqvae_ml32m.gin: 64): csrc/rq_generic.hip
        if not 1 <= n_layers <= _C.MAX_LEVELS:
            raise ValueError(f"n_layers must be in [1, {_C.MAX_LEVELS}]")
        self.input_dim, self.embed_dim, self.hidden_dims = input_dim, embed_dim, hidden_dims
        self.n_layers, self.codebook_size = n_layers, codebook_size
        self.commitment_weight, self.n_cat_feats = commitment_weight, n_cat_features
        self.codebook_normalize, self.codebook_mode = codebook_normalize, codebook_mode
        self.layers = nn.ModuleList([
            Quantize(embed_dim=embed_dim, n_embed=codebook_size, forward_mode=codebook_mode, do_kmeans_init=codebook_kmeans_init,
                     codebook_normalize=(i == 0 and codebook_normalize), sim_vq=codebook_sim_vq,
                     commitment_weight=commitment_weight) for i in range(n_layers)])
        self.encoder = MLP(input_dim=input_dim, hidden_dims=hidden_dims, out_dim=embed_dim, normalize=codebook_normalize)
        self.decoder = MLP(input_dim=embed_dim, hidden_dims=hidden_dims[-1::-1], out_dim=input_dim, normalize=True)
        self.reconstruction_loss = CategoricalReconstructionLoss(n_cat_features) if n_cat_features != 0 else ReconstructionLoss()  # rqvae.py:89-92
        # what HRqVae.forward consults; all inert here
        self.tag_alignment_weight = self.tag_prediction_weight = self.sem_id_uniqueness_weight = 0.0
        self.sem_id_uniqueness_loss = SemanticIdUniquenessLoss(margin=0.5, weight=0.0)
        self.tag_predictors, self.tag_projectors = nn.ModuleList(), nn.ModuleList()
        self.tag_class_counts, self.tag_embed_dim = [], 0
        self.rand = None

    def load_pretrained(self, path: str) -> None:
        from ..checkpoint import load_checkpoint
        state = load_checkpoint(path, map_location=self.device)
        self.load_state_dict(state["model"])
        print(f"---Loaded RQVAE Iter {state['iter']}---")

    def get_semantic_ids(self, x: Tensor, gumbel_t: float = 0.001) -> RqVaeOutput:  # takes the RAW input (rqvae.py:112-137)
        q = HRqVae.get_semantic_ids(self, self.encode(x), None, None, gumbel_t)
        return RqVaeOutput(embeddings=q.embeddings, residuals=q.residuals, sem_ids=q.sem_ids, quantize_loss=q.quantize_loss)

    def forward(self, batch, gumbel_t: float = 1.0) -> RqVaeComputedLosses:
        only_x = type("B", (), {})()
        only_x.x = batch.x
        out = HRqVae.forward(self, only_x, gumbel_t)
        s = self.last_summary  # device [6]: the total-loss launch already reduced the two means
        return RqVaeComputedLosses(loss=out.loss, reconstruction_loss=s[1], rqvae_loss=s[2], embs_norm=out.embs_norm,
                                   p_unique_ids=out.p_unique_ids)

    def predict_tags(self, *a, **k):
        raise AttributeError("RqVae has no tag heads")
