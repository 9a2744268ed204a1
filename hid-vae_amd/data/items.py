"""Item tensors resident in device memory.  The reference's `data.tags_processed.ItemData` wraps a torch_geometric
dataset built by a download + sentence-T5 pipeline (out of scope here); what the training loop consumes
is only x [N,768], tags_emb [N,T,768], tags_indices [N,T] and a train/eval split, which is what this class holds.

Accepted sources: a dict / .pt file with keys {x, tags_emb, tags_indices, is_train} (tags optional), or synthetic items."""
from enum import Enum

import torch

from .. import gin_compat as gin
from .schemas import SeqBatch, TaggedSeqBatch


@gin.constants_from_enum(module="data.tags_processed")
class RecDataset(Enum):
    """Dataset selector named in the reference configs (`%data.tags_processed.RecDataset.AMAZON`, reference
    data/tags_processed.py:20-25).  Here it only picks which resident item file to look for under `dataset_folder`."""
    AMAZON = 1
    ML_1M = 2
    ML_32M = 3
    KUAIRAND = 4



class _Opaque:
    """stands in for any torch_geometric class named inside the reference's processed file (only the tensors are wanted)"""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"state": state})


class _GeometricFreePickle:
    """pickle_module for torch.load that resolves `torch_geometric.*` globals to _Opaque, so the reference's processed dataset
    (a torch.save of (HeteroData.to_dict(), slices, <class HeteroData>)) loads without torch_geometric installed"""
    import pickle as _pickle

    __name__ = "pickle"
    Pickler = _pickle.Pickler
    load = staticmethod(_pickle.load)
    dump = staticmethod(_pickle.dump)

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] == "torch_geometric":
                return type(name, (_Opaque,), {})
            return super().find_class(module, name)


def load_reference_processed(path):
    """-> dict with x [N,768], optional tags_emb [N,T,768], tags_indices [N,T], is_train [N] from either file layout."""
    try:
        blob = torch.load(path, map_location="cpu", weights_only=False)
    except (ModuleNotFoundError, AttributeError, ImportError):
        blob = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_GeometricFreePickle)
    if isinstance(blob, dict) and "x" in blob:
        return blob
    store = None
    if isinstance(blob, (tuple, list)) and len(blob) >= 1 and isinstance(blob[0], dict):  # (data.to_dict(), slices, cls)
        store = blob[0].get("item")
    elif hasattr(blob, "__dict__"):  # a pickled HeteroData object
        stores = getattr(blob, "_node_store_dict", None) or {}
        store = stores.get("item")
    if store is not None and not isinstance(store, dict):
        store = getattr(store, "_mapping", None) or getattr(store, "__dict__", {}).get("_mapping")
    if not isinstance(store, dict) or "x" not in store:
        raise ValueError(f"{path}: neither a {{x, tags_emb, tags_indices, is_train}} dict nor the reference's processed HeteroData file")
    out = {"x": torch.as_tensor(store["x"])}
    for k in ("tags_emb", "tags_indices", "is_train"):
        if k in store and store[k] is not None:
            out[k] = torch.as_tensor(store[k])
    return out


class ResidentItemData:
    def __init__(self, x, tags_emb=None, tags_indices=None, device=None):
        dev = device or x.device
        self.x = x.to(dev).float().contiguous()
        self.tags_emb = tags_emb.to(dev).float().contiguous() if tags_emb is not None else None
        self.tags_indices = tags_indices.to(dev).long().contiguous() if tags_indices is not None else None

    @property
    def has_tags(self):
        return self.tags_emb is not None and self.tags_indices is not None

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, idx):
        """idx: int, slice or index tensor -> a (Tagged)SeqBatch whose `x` is [B,768] (items, not sequences)."""
        if isinstance(idx, int):
            idx = torch.tensor([idx], device=self.x.device)
        elif isinstance(idx, torch.Tensor):
            idx = idx.to(self.x.device)
        x = self.x[idx]
        ids = idx if isinstance(idx, torch.Tensor) else torch.arange(len(self), device=self.x.device)[idx]
        empty = torch.empty(0, device=self.x.device)
        mask = torch.ones(x.shape[0], dtype=torch.bool, device=self.x.device)
        if self.has_tags:
            return TaggedSeqBatch(user_ids=-torch.ones_like(ids), ids=ids, ids_fut=-torch.ones_like(ids), x=x, x_fut=empty,
                                  seq_mask=mask, tags_emb=self.tags_emb[idx], tags_indices=self.tags_indices[idx])
        return SeqBatch(user_ids=-torch.ones_like(ids), ids=ids, ids_fut=-torch.ones_like(ids), x=x, x_fut=empty, seq_mask=mask)

    def gather_into(self, idx, out):
        """rows `idx` (index tensor) written into the tensors of `out` (x, and tags_emb / tags_indices when present there): the batch
        lands where its consumer reads it -- GraphedTrainStep.input_buffers() -- instead of in a fresh tensor that is copied once more.
        -> out"""
        idx = idx.to(self.x.device)
        tagged = self.has_tags and getattr(out, "tags_emb", None) is not None
        if self.x.is_cuda:  # resident tables: one HIP launch for all of them (hidvae_gather_rows)
            from .. import _C
            tables = [self.x, self.tags_emb, self.tags_indices] if tagged else [self.x]
            outs = [out.x, out.tags_emb, out.tags_indices] if tagged else [out.x]
            if all(t.is_contiguous() and o.is_contiguous() and t.dtype == o.dtype for t, o in zip(tables, outs)):
                _C.gather_rows(idx.contiguous(), tables, outs)
                return out
        torch.index_select(self.x, 0, idx, out=out.x)
        if tagged:
            torch.index_select(self.tags_emb, 0, idx, out=out.tags_emb)
            torch.index_select(self.tags_indices, 0, idx, out=out.tags_indices)
        return out

    def subset(self, mask):
        return ResidentItemData(self.x[mask], self.tags_emb[mask] if self.has_tags else None,
                                self.tags_indices[mask] if self.has_tags else None)

    @staticmethod
    def from_file(path, device):
        """A plain dict {x, tags_emb, tags_indices, is_train}, or the reference's processed dataset file itself
        (`title_data_<split>_5tags.pt`, written by torch_geometric's InMemoryDataset.save in data/tags_amazon.py:392-431)."""
        blob = load_reference_processed(path)
        full = ResidentItemData(blob["x"], blob.get("tags_emb"), blob.get("tags_indices"), device=device)
        is_train = blob.get("is_train")
        return full, (is_train.to(device).bool() if is_train is not None else None)

    @staticmethod
    def synthetic(n_items, input_dim=768, n_tag_levels=3, class_counts=(38, 168, 348), tag_embed_dim=768, seed=0, device="cuda",
                  tagged=True):
        g = torch.Generator(device="cpu").manual_seed(seed)
        x = torch.nn.functional.normalize(torch.randn(n_items, input_dim, generator=g), dim=-1)
        te = ti = None
        if tagged:
            te = torch.randn(n_items, n_tag_levels, tag_embed_dim, generator=g)
            ti = torch.stack([torch.randint(0, c, (n_items,), generator=g) for c in class_counts[:n_tag_levels]], dim=1)
            ti[torch.rand(ti.shape, generator=g) < 0.05] = -1
        return ResidentItemData(x, te, ti, device=device)


class RandomBatches:
    """Endless stream of random batches drawn on the device from a per-rank seeded generator (the reference cycles a
    RandomSampler DataLoader, unseeded: train_hidvae.py:211-213)."""

    def __init__(self, data: ResidentItemData, batch_size: int, seed: int):
        self.data, self.batch_size = data, batch_size
        self.gen = torch.Generator(device=data.x.device).manual_seed(seed)
        self._perm, self._pos = None, 0

    def next(self, out=None):
        """the next batch; out: tensors to write it into (GraphedTrainStep.input_buffers(j)) when their shapes fit, else a fresh batch"""
        n = len(self.data)
        if self._perm is None or self._pos + self.batch_size > n:
            self._perm = torch.randperm(n, device=self.data.x.device, generator=self.gen)
            self._pos = 0
        idx = self._perm[self._pos:self._pos + min(self.batch_size, n)]
        self._pos += self.batch_size
        if out is not None and out.x.shape[0] == idx.shape[0]:
            return self.data.gather_into(idx, out)
        return self.data[idx]
