"""HRqVae on the MI355X-native path: same constructor / methods / state-dict keys as the reference
(modules/h_rqvae.py:230-756), but forward() is a short chain of fused HIP launches:

    x --[MFMA GEMM+SiLU x4]--> y --[fused RQ: l2norm, L x (search, rotation/STE, loss, residual)]--> z, ids, o_i, sum o_i
      --[tag heads on the concatenated o_i: projector, InfoNCE, predictor, focal+mixup]--
      --[MFMA GEMM+SiLU x4]--> y_dec --[l2norm + reconstruction loss]--> recon --[total loss + uniqueness term]--> loss

The reference's quirks that change results are reproduced on purpose and cited where they occur
(Q3 transposed uniqueness call, Q4 double weights, Q5 layer_idx=0, Q6 rotation residuals)."""
from typing import Dict, List, Optional

import os

import torch
from torch import Tensor, nn

from .. import _C
from ..data.schemas import HRqVaeComputedLosses, HRqVaeOutput
from ..ops import RQFn, StepLossFn, TotalLossFn  # noqa: F401
from .encoder import MLP
from .loss import CategoricalReconstructionLoss, QuantizeLoss, ReconstructionLoss, TagAlignmentLoss, TagPredictionLoss  # noqa: F401
from .quantize import Quantize, QuantizeForwardMode

try:  # the reference mixes in PyTorchModelHubMixin (h_rqvae.py:230); keep it when the hub package is importable
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:  # noqa: BLE001
    class _HubMixin:  # type: ignore
        pass


class SemanticIdUniquenessLoss(nn.Module):
    """Hinge on the cosine similarity of items that share a full semantic-id tuple (reference h_rqvae.py:25-105)."""

    def __init__(self, margin: float = 0.5, weight: float = 1.0):
        super().__init__()
        self.margin = margin
        self.weight = weight

    def forward(self, sem_ids: Tensor, encoded_features: Tensor) -> Tensor:
        # literal semantics for a [n, m] id matrix: rows are compared with each other (h_rqvae.py:52-64); differentiable with
        # respect to encoded_features like the reference's (the gradient lands on rows [0, n))
        from ..ops import UniqLossFn
        return UniqLossFn.apply(sem_ids.t().contiguous(), encoded_features, self.weight, self.margin)


class TagPredictor(nn.Module):
    """Gated residual MLP classifier over the concatenated code embeddings (reference h_rqvae.py:108-227)."""

    def __init__(self, embed_dim: int, num_classes: int, hidden_dim: Optional[int] = None, dropout_rate: float = 0.2,
                 use_batch_norm: bool = True, layer_idx: int = 0):
        super().__init__()
        hidden_dim = embed_dim * 2 if hidden_dim is None else hidden_dim
        p = min(0.55, dropout_rate + layer_idx * 0.075)
        mid = int(hidden_dim * 0.9)
        norm = (lambda n: nn.LayerNorm(n)) if use_batch_norm else (lambda n: nn.Identity())
        self.attention = nn.Sequential(nn.Linear(embed_dim, embed_dim // 4), nn.ReLU(), nn.Linear(embed_dim // 4, embed_dim // 2),
                                       nn.GELU(), nn.Linear(embed_dim // 2, embed_dim), nn.Sigmoid())
        self.feature_extractor = nn.Sequential(nn.Linear(embed_dim, hidden_dim), norm(hidden_dim), nn.ReLU(), nn.Dropout(p))

        def block():
            return nn.Sequential(nn.Linear(hidden_dim, mid), norm(mid), nn.ReLU(), nn.Dropout(p),
                                 nn.Linear(mid, hidden_dim), nn.ReLU(), nn.Dropout(p), norm(hidden_dim))

        self.residual_block1 = block()
        self.residual_block2 = block()
        self.classifier = nn.Sequential(nn.Linear(hidden_dim, mid), norm(mid), nn.ReLU(), nn.Dropout(p),
                                        nn.Linear(mid, mid // 2), nn.ReLU(), nn.Dropout(p * 0.5), nn.Linear(mid // 2, num_classes))
        self.label_smoothing = 0.1 if layer_idx > 0 else 0.05
        self.apply_norm = layer_idx > 0
        self.use_norm = use_batch_norm
        self.dropout_p = p
        self.rand = None

    def forward(self, x: Tensor) -> Tensor:
        from ..tagpath import tag_predictor_forward
        return tag_predictor_forward(self, x)


class HRqVae(nn.Module, _HubMixin):
    def __init__(self, input_dim: int, embed_dim: int, hidden_dims: List[int], codebook_size: int,
                 codebook_kmeans_init: bool = True, codebook_normalize: bool = False, codebook_sim_vq: bool = False,
                 codebook_mode: QuantizeForwardMode = QuantizeForwardMode.GUMBEL_SOFTMAX, n_layers: int = 3,
                 commitment_weight: float = 0.25, n_cat_features: int = 18, tag_alignment_weight: float = 0.5,
                 tag_prediction_weight: float = 0.5, tag_class_counts: Optional[List[int]] = None, tag_embed_dim: int = 768,
                 use_focal_loss: bool = False, focal_loss_params: Optional[Dict] = None, dropout_rate: float = 0.2,
                 use_batch_norm: bool = True, alignment_temperature: float = 0.1, sem_id_uniqueness_weight: float = 0.5,
                 sem_id_uniqueness_margin: float = 0.5) -> None:
        # the reference stores locals() here, `self` included; the module itself is left out on purpose
        self._config = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        super().__init__()
        if not 0 <= n_cat_features < input_dim:
            raise ValueError(f"n_cat_features={n_cat_features} must leave at least one of the {input_dim} input columns non-categorical")
        _C.check_embed_dim(embed_dim)  # 32: the fused kernels; other multiples of 4 up to 64 (rqvae_ml32m.gin: 64): csrc/rq_generic.hip
        if not 1 <= n_layers <= _C.MAX_LEVELS:
            raise ValueError(f"n_layers must be in [1, {_C.MAX_LEVELS}]")
        self.input_dim, self.embed_dim, self.hidden_dims = input_dim, embed_dim, hidden_dims
        self.n_layers, self.codebook_size = n_layers, codebook_size
        self.commitment_weight, self.n_cat_feats = commitment_weight, n_cat_features
        self.tag_alignment_weight, self.tag_prediction_weight = tag_alignment_weight, tag_prediction_weight
        self.tag_embed_dim, self.use_focal_loss = tag_embed_dim, use_focal_loss
        self.focal_loss_params = focal_loss_params or {"gamma": 2.0}
        self.dropout_rate, self.use_batch_norm = dropout_rate, use_batch_norm
        self.alignment_temperature = alignment_temperature
        self.sem_id_uniqueness_weight = sem_id_uniqueness_weight
        self.codebook_normalize, self.codebook_mode = codebook_normalize, codebook_mode
        self.tag_class_counts = ([10, 100, 1000] if tag_class_counts is None else list(tag_class_counts))[:n_layers]
        assert len(self.tag_class_counts) == n_layers, \
            f"Number of tag classes {len(self.tag_class_counts)} does not match number of layers {n_layers}"

        self.layers = nn.ModuleList([
            Quantize(embed_dim=embed_dim, n_embed=codebook_size, forward_mode=codebook_mode, do_kmeans_init=codebook_kmeans_init,
                     codebook_normalize=(i == 0 and codebook_normalize), sim_vq=codebook_sim_vq,
                     commitment_weight=commitment_weight) for i in range(n_layers)])
        self.concat_embed_dims = [embed_dim * (i + 1) for i in range(n_layers)]
        self._stored_tag_class_counts = None
        self.tag_predictors = nn.ModuleList([
            TagPredictor(embed_dim=self.concat_embed_dims[i], num_classes=self.tag_class_counts[i],
                         hidden_dim=hidden_dims[0] // 2 * (i + 1), dropout_rate=dropout_rate, use_batch_norm=use_batch_norm,
                         layer_idx=i) for i in range(n_layers)])
        self.tag_projectors = nn.ModuleList([self._make_projector(i, codebook_normalize) for i in range(n_layers)])
        self.encoder = MLP(input_dim=input_dim, hidden_dims=hidden_dims, out_dim=embed_dim, normalize=codebook_normalize)
        self.decoder = MLP(input_dim=embed_dim, hidden_dims=hidden_dims[-1::-1], out_dim=input_dim, normalize=True)
        self.reconstruction_loss = CategoricalReconstructionLoss(n_cat_features) if n_cat_features != 0 else ReconstructionLoss()  # h_rqvae.py:349-352
        self.tag_alignment_loss = TagAlignmentLoss(alignment_weight=tag_alignment_weight, temperature=alignment_temperature)
        self.tag_prediction_loss = TagPredictionLoss(use_focal_loss=use_focal_loss, focal_params=focal_loss_params, class_counts=None)
        self.sem_id_uniqueness_loss = SemanticIdUniquenessLoss(margin=sem_id_uniqueness_margin, weight=sem_id_uniqueness_weight)
        self.register_buffer("class_freq_counts", None)
        self.rand = None  # injectable randomness provider for dropout / mixup / gumbel (hidvae_amd.rand)
        self._census_tables = {}  # (user, B) -> id-census scratch; owned by the model so a captured HIP graph's addresses stay valid

    def _make_projector(self, i, with_layer_norm):
        h0 = self._config["hidden_dims"][0]
        return nn.Sequential(nn.Linear(self.tag_embed_dim, h0),
                             nn.BatchNorm1d(h0) if self._config["use_batch_norm"] else nn.Identity(), nn.ReLU(),
                             nn.Dropout(self._config["dropout_rate"]), nn.Linear(h0, self.concat_embed_dims[i]),
                             nn.LayerNorm(self.concat_embed_dims[i]) if with_layer_norm else nn.Identity())

    # ------------------------------------------------------------------------------------------ plumbing
    @property
    def config(self) -> dict:
        return self._config

    @property
    def device(self) -> torch.device:
        return next(self.encoder.parameters()).device

    def _zero_scalar(self, device):
        z = getattr(self, "_zero_cache", None)
        if z is None or z.device != device:
            z = self._zero_cache = torch.zeros((), device=device)  # one fill for the model's lifetime, not per step
        return z

    def _census(self, user, B, device):
        """the model's own id-census table for `user` ('fused' middle launch / 'stats' launch) at batch size B.  Never evicted: a
        captured graph holds its address.  (One table per user: the two kernels lay their slots out differently.)"""
        tabs = self.__dict__.setdefault("_census_tables", {})
        key = (user, int(B), str(device))
        t = tabs.get(key)
        if t is None:
            t = tabs[key] = _C.census_scratch(B, device)
        return t

    def _cut_here(self, t):
        """identity, or -- while a split backward is being recorded -- a fresh leaf standing in for `t`: the first backward pass
        stops there (leaving d loss / d t in the leaf's .grad), backward_rest() resumes from `t` with that gradient"""
        if not getattr(self, "_cutting", False) or t is None or not t.requires_grad:
            return t
        leaf = t.detach().requires_grad_()
        self._cut_pairs.append((t, leaf))
        return leaf

    def backward_rest(self):
        """second half of a split backward: everything upstream of the cuts of the last forward, in ONE autograd pass"""
        pairs = [(t, leaf.grad) for t, leaf in (self._cut_pairs or []) if leaf.grad is not None]
        self._cut_pairs = None
        if pairs:
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])

    def dp_first_bucket(self, batch_size):
        """the parameters whose gradients the FIRST half of a split backward completes (the decoder's last two layers when the fused
        middle launch runs, else the whole decoder): data-parallel runs put them first in the flat gradient buffer"""
        Wd = self.decoder.weights()
        We = self.encoder.weights()
        fused = (len(We) >= 3 and len(Wd) >= 3 and self.codebook_mode.value in (QuantizeForwardMode.STE.value, QuantizeForwardMode.ROTATION_TRICK.value)
                 and getattr(self, "fuse_bottleneck", True) and self.embed_dim == _C.EMBED_DIM
                 and _C.bottleneck_eligible(batch_size, We[-2].shape[1], We[-2].shape[0], Wd[0].shape[0], Wd[1].shape[0], self.n_layers, self.codebook_size))
        return list(Wd[2:]) if fused else list(Wd)

    def _rand(self):
        if self.rand is not None:
            return self.rand
        from ..rand import DeviceRand
        alpha = self.tag_prediction_loss.mixup_alpha
        r = getattr(self, "_device_rand", None)
        if r is None or r.mixup_alpha != alpha:
            r = self._device_rand = DeviceRand(alpha)  # kept: it caches device-side distribution objects
        return r

    def load_pretrained(self, path: str) -> None:
        """Checkpoint loader tolerant of tag-head shape drift (reference h_rqvae.py:382-471)."""
        from ..checkpoint import load_checkpoint
        state = load_checkpoint(path, map_location=self.device)
        theirs, mine = state["model"], self.state_dict()
        classes = []
        for i in range(self.n_layers):
            key = f"tag_predictors.{i}.classifier.7.weight"
            classes.append(theirs[key].shape[0] if key in theirs else self.tag_class_counts[i])
        if classes != list(self.tag_class_counts):
            print(f"Tag predictor mismatch detected. Adjusting number of classes from {self.tag_class_counts} to {classes}")
            self._stored_tag_class_counts, self.tag_class_counts = self.tag_class_counts, classes
            self.tag_predictors = nn.ModuleList([
                TagPredictor(embed_dim=self.concat_embed_dims[i], num_classes=classes[i],
                             hidden_dim=self._config.get("hidden_dims", [512, 256, 128])[0] // 2 * (i + 1),
                             dropout_rate=self._config.get("dropout_rate", 0.2),
                             use_batch_norm=self._config.get("use_batch_norm", True), layer_idx=i)
                for i in range(self.n_layers)]).to(self.device)
        if any(f"tag_projectors.{i}.5.weight" in theirs and f"tag_projectors.{i}.5.weight" not in mine for i in range(self.n_layers)):
            print("Tag projector mismatch detected. Adjusting structure to match the weight file.")
            self.tag_projectors = nn.ModuleList([self._make_projector(i, True) for i in range(self.n_layers)]).to(self.device)
        mine = self.state_dict()
        usable = {k: v for k, v in theirs.items() if k in mine}
        if len(usable) < len(theirs):
            print(f"Warning: skipped keys absent from this model: {sorted(set(theirs) - set(usable))}")
        try:
            mine.update(usable)
            self.load_state_dict(mine)
        except Exception as e:  # noqa: BLE001  same last resort as the reference
            print(f"Standard loading failed, trying strict=False: {e}")
            self.load_state_dict(theirs, strict=False)
        for layer in self.layers:
            layer.kmeans_initted = True
        print(f"---Loaded HRQVAE Iter {state['iter']}---")

    def update_class_counts(self, class_counts_dict):
        # inert for the loss, exactly as in the reference (h_rqvae.py:740-756)
        for layer_idx, counts in class_counts_dict.items():
            if not isinstance(counts, torch.Tensor):
                counts = torch.tensor(counts, device=self.device)
            self.register_buffer(f"class_freq_counts_{layer_idx}", counts)
        self.class_freq_layers = list(class_counts_dict.keys())

    # ------------------------------------------------------------------------------------------ pieces
    def encode(self, x: Tensor) -> Tensor:
        return self.encoder(x.float())

    def decode(self, x: Tensor) -> Tensor:
        return self.decoder(x)

    def _tables(self):
        return [layer.table() for layer in self.layers]

    def _normalize_flags(self):
        return tuple(layer.codebook_normalize for layer in self.layers)

    def _maybe_kmeans(self, y, normalize_input):
        """First-batch k-means of every level on that level's input residual (reference quantize.py:103-104 fires
        inside the level loop; levels are initialised in order, each on the residual the previous ones leave)."""
        pending = [i for i, layer in enumerate(self.layers) if layer.do_kmeans_init and not layer.kmeans_initted]
        if not pending:
            return
        with torch.no_grad():
            for i in pending:
                tabs = [t.detach() for t in self._tables()[: i + 1]]
                out = RQFn.apply(y.detach(), normalize_input, self._fused_mode(), self.training, self.commitment_weight,
                                 self._normalize_flags()[: i + 1], True, None, None, *tabs)
                res = out[5][:, i * self.embed_dim:(i + 1) * self.embed_dim].contiguous()
                self.layers[i]._kmeans_init(res)

    def _prepare_codebooks_async(self):
        if any(layer.do_kmeans_init and not layer.kmeans_initted for layer in self.layers):
            return None  # tables are about to change
        from ..ops import side_stream
        main, side = torch.cuda.current_stream(), side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            tabs = [t.detach() for t in self._tables()] if not self.layers[0].sim_vq else None
            if tabs is None:
                return None
            cb, cc = _C.codebook_prepare(tabs, self._normalize_flags())
        cb.record_stream(main)
        cc.record_stream(main)
        return cb, cc

    def _bottleneck_ok(self, x):
        """may this training step use the fused middle launch? (see include/hidvae.h, hidvae_bottleneck_fwd)"""
        if not (self.training and torch.is_grad_enabled() and getattr(self, "fuse_bottleneck", True)):
            return False
        if self.codebook_mode.value not in (QuantizeForwardMode.STE.value, QuantizeForwardMode.ROTATION_TRICK.value):
            return False
        if any(getattr(layer, "do_kmeans_init", False) and not getattr(layer, "kmeans_initted", True) for layer in self.layers):
            return False  # the k-means start-up pass wants y on its own first
        We, Wd = self.encoder.weights(), self.decoder.weights()
        if len(We) < 3 or len(Wd) < 3 or x.dim() != 2 or self.embed_dim != _C.EMBED_DIM:
            return False
        return _C.bottleneck_eligible(x.shape[0], We[-2].shape[1], We[-2].shape[0], Wd[0].shape[0], Wd[1].shape[0], self.n_layers,
                                      self.codebook_size)

    def _fused_mode(self):
        m = self.codebook_mode.value
        return _C.MODE_STE if m == QuantizeForwardMode.GUMBEL_SOFTMAX.value else m

    def _quantize_all(self, y, normalize_input, want_res):
        """-> z, ids [B,L], emb_cat [B,L*D], emb_sum [B,D], qloss [B], res_cat"""
        self._maybe_kmeans(y, normalize_input)
        if self.training and self.codebook_mode == QuantizeForwardMode.GUMBEL_SOFTMAX:
            from ..gumbel_path import gumbel_all_levels
            return gumbel_all_levels(self, y, normalize_input)
        return RQFn.apply(y.contiguous(), normalize_input, self._fused_mode(), self.training, self.commitment_weight,
                          self._normalize_flags(), want_res, getattr(self, "_prepared", None), getattr(self, "_heads_port", None),
                          *self._tables())

    def _tag_heads(self, emb_cat, tags_emb, tags_indices, defer_join=False, port=None, loss_grad=None, projs=None):
        from ..tagpath import tag_heads_forward
        return tag_heads_forward(self, emb_cat, tags_emb, tags_indices, defer_join=defer_join, port=port, loss_grad=loss_grad, projs=projs)

    def get_semantic_ids(self, encoded_x: Tensor, tags_emb: Optional[Tensor] = None, tags_indices: Optional[Tensor] = None,
                         gumbel_t: float = 0.001) -> HRqVaeOutput:
        self._gumbel_t = gumbel_t
        z, ids, emb_cat, emb_sum, qloss, res_cat = self._quantize_all(encoded_x.float().contiguous(), False, True)
        B, L, D = z.shape[0], self.n_layers, self.embed_dim
        zero = self._zero_scalar(z.device)
        align = pred = acc = zero
        by_layer = (None, None, None) if tags_emb is not None and tags_indices is not None else ([], [], [])
        if tags_emb is not None and tags_indices is not None:
            sc = self._tag_heads(emb_cat, tags_emb.float(), tags_indices)
            n = len(sc) // 3
            by_layer = (torch.stack(sc[:n]), torch.stack(sc[n:2 * n]), torch.stack(sc[2 * n:]))  # (API path, not the hot step)
            align, pred, acc = by_layer[0].sum() / self.n_layers, by_layer[1].sum() / self.n_layers, by_layer[2].sum() / self.n_layers
        return HRqVaeOutput(embeddings=emb_cat.view(B, L, D).transpose(1, 2), residuals=res_cat.view(B, L, D).transpose(1, 2),
                            sem_ids=ids, quantize_loss=qloss, tag_align_loss=align, tag_pred_loss=pred, tag_pred_accuracy=acc,
                            tag_align_loss_by_layer=by_layer[0], tag_pred_loss_by_layer=by_layer[1],
                            tag_pred_accuracy_by_layer=by_layer[2])

    @torch.no_grad()
    def semantic_ids_only(self, encoded_x: Tensor) -> Tensor:
        """sem_ids [B, L] of get_semantic_ids(encoded_x) in eval mode and nothing else: the corpus pass of the tokenizer
        (reference h_semids.py:109-195 keeps only `.sem_ids` of the level loop's output).  One launch that writes 8 L bytes per item."""
        if self.training:
            raise RuntimeError("semantic_ids_only is the eval-mode search (model.eval() first); training outputs differ (SURVEY Q6)")
        y = encoded_x.float().contiguous()
        self._maybe_kmeans(y, False)
        if self.layers[0].sim_vq:
            return self.get_semantic_ids(y).sem_ids
        cb, cc = _C.codebook_prepare([t.detach() for t in self._tables()], self._normalize_flags())
        return _C.rq_ids(y, cb, cc, False)

    # ------------------------------------------------------------------------------------------ the step
    def forward(self, batch, gumbel_t: float = 1.0) -> HRqVaeComputedLosses:
        x = batch.x.float().contiguous()
        tags_emb = getattr(batch, "tags_emb", None)
        tags_indices = getattr(batch, "tags_indices", None)
        tagged = tags_emb is not None and tags_indices is not None
        self._gumbel_t = gumbel_t

        if self.training and tagged:
            r = self._rand()
            if hasattr(r, "begin_step"):
                # the generator's step counter and the mixup pairing (rand.DeviceRand) -- on the first tag stream, beside the encoder,
                # when the heads run on streams of their own (tagpath.early_rand)
                from ..tagpath import early_rand
                lm = self.tag_prediction_loss
                mix = torch.is_grad_enabled() and lm.use_mixup and x.shape[0] > 1 and hasattr(r, "prepare_mixup")
                if not early_rand(r, tags_indices[:, :self.n_layers], x.device, self.n_layers, mix):
                    r.begin_step(x.device)
        _C.phase_mark("fwd:start")
        self._prepared = self._prepare_codebooks_async()  # effective codebooks + |c|^2 on the helper stream, beside the encoder
        # data-parallel overlap (step.GraphedTrainStep): with `dp_cut` set the backward is split in two at the inputs of the decoder's
        # tail and of the loss launch, so the gradients of the decoder's last layers -- final first -- go on the wire while the rest of
        # the backward still runs (see _cut_here / backward_rest)
        self._cutting = bool(getattr(self, "dp_cut", False)) and self.training and torch.is_grad_enabled()
        self._cut_pairs = [] if self._cutting else None
        # loss_grad_hint (set by the training loop: the gradient it will hand loss.backward, 1 / gradient_accumulate_every): the tag heads
        # then run their backward right after their forward, on their own streams (tagpath.HeadsGradPort); None = everything waits for
        # loss.backward().  The hint is checked against the real gradient inside the loss backward launch (a mismatch poisons the step).
        hint = getattr(self, "loss_grad_hint", None)
        early_heads = (tagged and hint is not None and self.training and torch.is_grad_enabled() and x.dim() == 2
                       and self.codebook_mode != QuantizeForwardMode.GUMBEL_SOFTMAX)  # (the Gumbel path differentiates emb_cat through autograd)
        if early_heads:
            from ..tagpath import HeadsGradPort
            self._heads_port = HeadsGradPort((x.shape[0], self.n_layers * self.embed_dim))
        else:
            self._heads_port = None
        early_projs = None
        # the projectors need only the batch: they run on the level streams beside the front of the step (tagpath.tag_projectors_early) --
        # from the fused middle launch on, not from the step's first launch: beside the encoder's two wide layers they slow those down
        # (22 vs 13 us for the first), and the encoder is in front of EVERY lane; the middle launch runs on 64 CUs and leaves them room
        # (tagged step 1.049-1.058 -> 1.029-1.044 ms, B = 2048 1.726 -> 1.695; HIDVAE_PROJ_AFTER_ENC=0: from the first launch on)
        late_proj = self._bottleneck_ok(x) and os.environ.get("HIDVAE_PROJ_AFTER_ENC", "1") != "0"
        if early_heads and not late_proj:
            from ..tagpath import tag_projectors_early
            early_projs = tag_projectors_early(self, tags_emb.float(), self._rand())
        y_dec = None
        embs_norm = p_unique = None  # (the fused middle launch produces them itself when it can)
        fused = self._bottleneck_ok(x)
        if fused:
            # small batches: encoder[-2:] + the L levels + decoder[:2] are one launch (ops.BottleneckFn); the stacks either side
            # hand over (pre-activation, activation) pairs so no elementwise launch appears at the cuts
            from ..ops import BottleneckFn, MLPBackFn, MLPFrontFn
            We, Wd = self.encoder.weights(), self.decoder.weights()
            pre1, h1 = MLPFrontFn.apply(x, *We[:-2])
            if early_heads and late_proj:
                from ..tagpath import tag_projectors_early
                early_projs = tag_projectors_early(self, tags_emb.float(), self._rand())
            z, ids, emb_cat, emb_sum, qloss, pre_d1, d1, embs_norm, p_unique = BottleneckFn.apply(
                pre1, h1, We[-2], We[-1], Wd[0], Wd[1], self.codebook_normalize, self._fused_mode(), self.commitment_weight,
                self._normalize_flags(), self._prepared, (lambda: self._census("fused", x.shape[0], x.device)), self._heads_port,
                *self._tables())
        else:
            y = self.encoder.body(x)  # the encoder's l2norm (codebook_normalize) happens in the RQ prologue
            z, ids, emb_cat, emb_sum, qloss, _ = self._quantize_all(y, self.codebook_normalize, False)
        self._prepared = None
        _C.phase_mark("fwd:quantised")

        # the tag heads need only emb_cat: their per-level branches fork off HERE, on streams of their own, and the decoder is issued
        # on the caller's stream beside them (round 2 issued the decoder first and made the branches wait for it); the join is
        # deferred to just before the loss launch
        tag_scalars, tag_join = (), None
        if tagged:
            tag_scalars, tag_join = self._tag_heads(emb_cat, tags_emb.float(), tags_indices, defer_join=True, port=self._heads_port,
                                                    loss_grad=hint if early_heads else None, projs=early_projs)  # (A_0.., P_0.., acc_0..)
        self._heads_port = None
        if fused:
            dec_in = self._cut_here(pre_d1)
            y_dec = MLPBackFn.apply(dec_in, d1, *Wd[2:])
        else:
            y_dec = self.decoder.body(self._cut_here(emb_sum))
        _C.phase_mark("fwd:decoder done")
        if tag_join is not None:
            tag_join()
        _C.phase_mark("fwd:heads joined")
        if getattr(self, "_cutting", False):  # everything else the loss launch differentiates is cut too (see _cut_here)
            n_t = len(tag_scalars) // 3
            qloss, z = self._cut_here(qloss), self._cut_here(z)
            tag_scalars = tuple(self._cut_here(t) for t in tag_scalars[:2 * n_t]) + tuple(tag_scalars[2 * n_t:])

        # debug statistics of h_rqvae.py:643-648 (embs_norm, p_unique_ids): ready as soon as the ids are, so they run on the
        # helper stream beside the decoder instead of after it
        from ..ops import side_stream
        main, side = torch.cuda.current_stream(), side_stream()
        if embs_norm is None:
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                embs_norm, p_unique = _C.id_stats(emb_cat.detach(), ids, scratch=self._census("stats", ids.shape[0], ids.device))
            for t in (embs_norm, p_unique):
                t.record_stream(main)
            for t in (emb_cat, ids):
                t.record_stream(side)
        # decoder l2norm + sum (x_hat-x)^2 (Q7; with n_cat > 0 the head is normalised once more and the last n_cat columns enter as
        # BCE-with-logits, h_rqvae.py:610-613 + loss.py:15-33) and the total loss in one launch
        # the alignment / uniqueness weights enter once inside their loss modules and once more here
        n_tag = len(tag_scalars) // 3
        loss, recon, uniq, stats, summary = StepLossFn.apply(y_dec, x, qloss, z, ids, self.sem_id_uniqueness_loss.weight,
                                                    self.sem_id_uniqueness_loss.margin, self.tag_alignment_weight,
                                                    self.tag_prediction_weight, self.sem_id_uniqueness_weight, n_tag,
                                                    float(self.n_layers),
                                                    (int(self.n_cat_feats), float(hint)) if early_heads else int(self.n_cat_feats), *tag_scalars)
        main.wait_stream(side)  # the statistics above were computed beside the decoder
        _C.phase_mark("fwd:loss done")
        self.last_summary = summary  # device [6]: loss, mean recon, mean rqvae, tag align, tag pred, tag accuracy (training log row)
        zero = self._zero_scalar(x.device)
        if tagged:
            L = n_tag
            return HRqVaeComputedLosses(
                loss=loss, reconstruction_loss=recon, rqvae_loss=qloss, tag_align_loss=stats[0], tag_pred_loss=stats[1],
                tag_pred_accuracy=stats[2], embs_norm=embs_norm, p_unique_ids=p_unique, tag_align_loss_by_layer=stats[3:3 + L],
                tag_pred_loss_by_layer=stats[3 + L:3 + 2 * L], tag_pred_accuracy_by_layer=stats[3 + 2 * L:3 + 3 * L],
                sem_id_uniqueness_loss=uniq)
        return HRqVaeComputedLosses(
            loss=loss, reconstruction_loss=recon, rqvae_loss=qloss, tag_align_loss=zero, tag_pred_loss=zero, tag_pred_accuracy=zero,
            embs_norm=embs_norm, p_unique_ids=p_unique, tag_align_loss_by_layer=[], tag_pred_loss_by_layer=[],
            tag_pred_accuracy_by_layer=[], sem_id_uniqueness_loss=uniq)

    @torch.no_grad()
    def predict_tags(self, x: Tensor, gumbel_t: float = 0.001) -> Dict[str, Tensor]:
        """argmax tag + confidence per level (reference h_rqvae.py:674-738)."""
        shape = x.shape
        flat = x.reshape(-1, shape[-1]) if x.dim() == 3 else x
        was_training = self.training
        z = self.encode(flat)
        out = self.get_semantic_ids(z, None, None, gumbel_t)
        emb_cat = out.embeddings.transpose(1, 2).reshape(z.shape[0], -1)
        n = z.shape[0]
        preds = torch.empty((n, self.n_layers), dtype=torch.int64, device=z.device)
        confs = torch.empty((n, self.n_layers), dtype=torch.float32, device=z.device)
        for i in range(self.n_layers):  # the eval-mode launch chain of the heads, then ONE launch for arg max + its softmax probability
            logits = self.tag_predictors[i](emb_cat[:, : (i + 1) * self.embed_dim])
            _C.softmax_argmax_rows(logits.contiguous(), preds, confs, i)
        self.train(was_training)
        if x.dim() == 3:
            preds, confs = preds.reshape(shape[0], shape[1], self.n_layers), confs.reshape(shape[0], shape[1], self.n_layers)
        return {"predictions": preds, "confidences": confs}
