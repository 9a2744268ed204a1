"""HSemanticIdTokenizer on the HIP path (reference modules/tokenizer/h_semids.py:24-532): owns an eval-mode HRqVae,
turns item features into semantic-id tuples (optionally concatenated / interleaved with predicted tag ids), caches the
corpus ids and answers prefix-existence queries for constrained decoding.

Differences in HOW (not what): the corpus is encoded in large resident chunks through the fused encode + RQ kernels
instead of 512-item DataLoader batches, and `exists_prefix` is a sorted-key binary search (one sort per prefix length,
cached) instead of a [queries, corpus, L] broadcast compare."""
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from ...data.schemas import SeqBatch, TokenizedSeqBatch
from ..h_rqvae import HRqVae

BATCH_SIZE = 16  # the reference checks prefixes in groups of 16 (h_semids.py:22,218)
CORPUS_CHUNK = 65536


def _eval_mode(fn):
    def inner(self, *args, **kwargs):
        was = self.training
        self.eval()
        try:
            return fn(self, *args, **kwargs)
        finally:
            self.train(was)
    return inner


class HSemanticIdTokenizer(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, hidden_dims: List[int], codebook_size: int, n_layers: int = 3,
                 n_cat_feats: int = 18, commitment_weight: float = 0.25, hrqvae_weights_path: Optional[str] = None,
                 hrqvae_codebook_normalize: bool = False, hrqvae_sim_vq: bool = False, tag_alignment_weight: float = 0.5,
                 tag_prediction_weight: float = 0.5, tag_class_counts: Optional[List[int]] = None, tag_embed_dim: int = 768,
                 use_dedup_dim: bool = False, use_concatenated_ids: bool = False, use_interleaved_ids: bool = False) -> None:
        super().__init__()
        if sum(map(bool, (use_dedup_dim, use_concatenated_ids, use_interleaved_ids))) > 1:
            raise ValueError("use_dedup_dim, use_concatenated_ids and use_interleaved_ids are mutually exclusive")
        self.hrq_vae = HRqVae(input_dim=input_dim, embed_dim=output_dim, hidden_dims=hidden_dims, codebook_size=codebook_size,
                              codebook_kmeans_init=False, codebook_normalize=hrqvae_codebook_normalize,
                              codebook_sim_vq=hrqvae_sim_vq, n_layers=n_layers, n_cat_features=n_cat_feats,
                              commitment_weight=commitment_weight, tag_alignment_weight=tag_alignment_weight,
                              tag_prediction_weight=tag_prediction_weight, tag_class_counts=tag_class_counts,
                              tag_embed_dim=tag_embed_dim)
        if hrqvae_weights_path is not None:
            self.hrq_vae.load_pretrained(hrqvae_weights_path)
        self.hrq_vae.eval()
        self.codebook_size, self.n_layers = codebook_size, n_layers
        self.use_dedup_dim, self.use_concatenated_ids, self.use_interleaved_ids = use_dedup_dim, use_concatenated_ids, use_interleaved_ids
        self.tag_class_counts = tag_class_counts
        self.reset()

    def reset(self):
        self.cached_ids = None

    @property
    def cached_ids(self):
        return self.__dict__.get("_cached_ids")

    @cached_ids.setter
    def cached_ids(self, ids):
        # (an assignment from outside drops the sorted-prefix index and the knowledge that every id is below the codebook size)
        self.__dict__["_cached_ids"] = ids
        self._prefix_index = {}
        self._ids_trusted = False

    @property
    def sem_ids_dim(self):
        if self.use_dedup_dim:
            return self.n_layers + 1
        if (self.use_concatenated_ids or self.use_interleaved_ids) and self.tag_class_counts is not None:
            return self.n_layers + len(self.tag_class_counts)
        return self.n_layers

    # ------------------------------------------------------------------------------------------------
    def _ids_for(self, feats: Tensor) -> Tensor:
        """feats [..., input_dim] -> ids [n_items, sem_ids_dim-ish] (semantic ids, plus tag ids in the combined modes)."""
        flat = feats.reshape(-1, feats.shape[-1]).to(self.hrq_vae.device)
        if not (self.use_concatenated_ids or self.use_interleaved_ids):
            if self.hrq_vae.training:
                # (a parent module's .train() reached the tokenizer's model: the reference calls get_semantic_ids whatever the mode,
                #  h_semids.py:128,271; the ids-only launch is the eval-mode search)
                return self.hrq_vae.get_semantic_ids(self.hrq_vae.encode(flat)).sem_ids
            return self.hrq_vae.semantic_ids_only(self.hrq_vae.encode(flat))  # only the ids leave the launch
        sem = self.hrq_vae.get_semantic_ids(self.hrq_vae.encode(flat)).sem_ids
        tags = self.hrq_vae.predict_tags(flat)["predictions"]
        if tags.shape[0] != sem.shape[0]:
            raise ValueError(f"Semantic ID batch size ({sem.shape[0]}) does not match predicted tag batch size ({tags.shape[0]})")
        if self.use_concatenated_ids:
            return torch.cat([sem, tags], dim=1)
        cols = []  # interleave s1,t1,s2,t2,... (h_semids.py:160-170)
        for i in range(max(sem.shape[1], tags.shape[1])):
            if i < sem.shape[1]:
                cols.append(sem[:, i:i + 1])
            if i < tags.shape[1]:
                cols.append(tags[:, i:i + 1])
        return torch.cat(cols, dim=1)

    @staticmethod
    def _corpus_features(dataset) -> Tensor:
        if isinstance(dataset, Tensor):
            return dataset
        x = getattr(dataset, "x", None)
        if isinstance(x, Tensor):
            return x
        if hasattr(dataset, "__len__") and hasattr(dataset, "__getitem__"):
            n = len(dataset)
            if n == 0:
                return torch.empty(0, 0)
            batch = dataset[torch.arange(n)] if not isinstance(dataset, (list, tuple)) else None
            if batch is not None and hasattr(batch, "x"):
                return batch.x
            return torch.stack([(dataset[i].x if hasattr(dataset[i], "x") else dataset[i]) for i in range(n)])
        raise TypeError("precompute_corpus_ids: pass a feature tensor, an object with .x, or an indexable item dataset")

    @torch.no_grad()
    @_eval_mode
    def precompute_corpus_ids(self, movie_dataset) -> Tensor:
        feats = self._corpus_features(movie_dataset)
        if feats.numel() == 0:
            self.cached_ids = torch.empty(0, self.sem_ids_dim, device=self.hrq_vae.device, dtype=torch.long)
        else:
            parts = [self._ids_for(feats[i:i + CORPUS_CHUNK]) for i in range(0, feats.shape[0], CORPUS_CHUNK)]
            self.cached_ids = torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]
        self._prefix_index = {}
        self._ids_trusted = True
        return self.cached_ids

    # ------------------------------------------------------------------------------------------------
    def _keys(self, ids: Tensor, width: int, radix: int) -> Tensor:
        key = torch.zeros(ids.shape[:-1], dtype=torch.int64, device=ids.device)
        for j in range(width):
            key = key * radix + ids[..., j].to(torch.int64)
        return key

    @torch.no_grad()
    @_eval_mode
    def exists_prefix(self, sem_id_prefix: Tensor) -> Tensor:
        if self.cached_ids is None:
            raise Exception("No match found in empty cache.")
        width = min(sem_id_prefix.shape[-1], self.cached_ids.shape[-1])
        out = torch.zeros(*sem_id_prefix.shape[:-1], dtype=torch.bool, device=sem_id_prefix.device)
        if self.cached_ids.shape[0] == 0 or width == 0:
            return out
        if width not in self._prefix_index:
            # ids produced by precompute_corpus_ids are < codebook_size (semantic) / < the level's class count (predicted tags): the
            # radix is known without reading the device; a cache assigned from outside is checked once (one host read)
            bound = max(self.codebook_size - 1, *(self.tag_class_counts or [0]))
            if not getattr(self, "_ids_trusted", False):
                bound = max(bound, int(self.cached_ids.max()))
            radix = int(bound) + 2
            if radix ** width >= 2 ** 62:
                raise OverflowError("id prefix does not fit a 64-bit key")
            self._prefix_index[width] = (radix, torch.sort(self._keys(self.cached_ids[:, :width], width, radix)).values)
        radix, sorted_keys = self._prefix_index[width]
        q = sem_id_prefix[..., :width].to(self.cached_ids.device)
        ok = (q >= 0).all(dim=-1) & (q < radix).all(dim=-1)
        qk = self._keys(q.clamp(min=0, max=radix - 1), width, radix)
        pos = torch.searchsorted(sorted_keys, qk).clamp(max=sorted_keys.numel() - 1)
        hit = ((sorted_keys[pos] == qk) & ok).to(out.device)
        # the reference walks the rows in floor(rows/16) groups of 16 (h_semids.py:218: math.ceil(B // BATCH_SIZE)), so the
        # trailing rows % 16 rows are never examined and stay False -- reproduced
        covered = (sem_id_prefix.shape[0] // BATCH_SIZE) * BATCH_SIZE
        out[:covered] = hit[:covered]
        return out

    # ------------------------------------------------------------------------------------------------
    def _tokenize_seq_batch_from_cached(self, ids: Tensor) -> Tensor:
        valid = ids.clone()
        valid[valid >= self.cached_ids.shape[0]] = 0
        return self.cached_ids[valid.flatten(), :].reshape(ids.shape[0], -1)

    @torch.no_grad()
    @_eval_mode
    def forward(self, batch: SeqBatch) -> TokenizedSeqBatch:
        B, N = batch.ids.shape
        if self.cached_ids is None or batch.ids.max() >= self.cached_ids.shape[0]:
            ids = self._ids_for(batch.x)  # [B*N, D_total]
            D_total = ids.shape[1]
            sem_ids = ids.reshape(B, N * D_total)
            sem_ids_fut = None
            if batch.x_fut is not None:
                sem_ids_fut = self._ids_for(batch.x_fut.unsqueeze(1)).reshape(B, -1)
            seq_mask = batch.seq_mask.repeat_interleave(D_total, dim=1) if batch.seq_mask is not None else None
            if seq_mask is not None:
                sem_ids[~seq_mask] = -1
        else:
            D_total = self.cached_ids.shape[-1]
            sem_ids = self._tokenize_seq_batch_from_cached(batch.ids)
            seq_mask = batch.seq_mask.repeat_interleave(D_total, dim=1) if batch.seq_mask is not None else None
            if seq_mask is not None:
                sem_ids[~seq_mask] = -1
            sem_ids_fut = self._tokenize_seq_batch_from_cached(batch.ids_fut)
        ttype = torch.arange(D_total, device=sem_ids.device)
        return TokenizedSeqBatch(user_ids=batch.user_ids, sem_ids=sem_ids, sem_ids_fut=sem_ids_fut, seq_mask=seq_mask,
                                 token_type_ids=ttype.repeat(B, N), token_type_ids_fut=ttype.repeat(B, 1))

    @torch.no_grad()
    @_eval_mode
    def predict_tags(self, batch: SeqBatch) -> Dict[str, Tensor]:
        """Tag predictions with padded sequence positions masked to -1 / 0.0 (h_semids.py:453-515)."""
        seq_mask = getattr(batch, "seq_mask", None)
        if seq_mask is None:
            return self.hrq_vae.predict_tags(batch.x)
        x = batch.x * seq_mask.unsqueeze(-1).to(batch.x.dtype)
        pred = self.hrq_vae.predict_tags(x)
        m = seq_mask.unsqueeze(-1)
        pred["predictions"] = torch.where(m.expand_as(pred["predictions"]), pred["predictions"], torch.full_like(pred["predictions"], -1))
        pred["confidences"] = torch.where(m.expand_as(pred["confidences"]), pred["confidences"], torch.zeros_like(pred["confidences"]))
        return pred

    @torch.no_grad()
    @_eval_mode
    def tokenize_with_tags(self, batch: SeqBatch) -> Tuple[TokenizedSeqBatch, Dict[str, Tensor]]:
        return self.forward(batch), self.predict_tags(batch)
