#!/usr/bin/env python3
"""Generate tests/golden/tokenizer_*.npz and tests/golden/state_dict_contract.json by running the REFERENCE's
HSemanticIdTokenizer / HRqVae (/root/reference) in this container.

    TORCHDYNAMO_DISABLE=1 python tests/golden/make_golden_tokenizer.py

Same in-process shim as make_golden.py (stub `gin`, the loss-class typo alias) plus one more stand-in: the tokenizer module
imports `data.tags_processed` only for three names it uses as annotations (ItemData, SeqData, RecDataset); that module pulls in
the dataset ingestion stack (torch_geometric, polars: not installed), so a three-name placeholder module takes its place.  No
reference file is edited.  Weights and items are oracle.fill formulas; a fixture holds the reference's outputs only:
  * corpus ids from precompute_corpus_ids in the three id modes (plain / concatenated / interleaved with predicted tag ids),
    the corpus served as the reference serves it -- a torch DataLoader over SeqBatch records, batch_size 512 (h_semids.py:120);
  * exists_prefix truth tables for 2-D prefixes of every width and for a 3-D prefix, incl. the rows the reference never examines
    (`ceil(n // 16)` batches, h_semids.py:218);
  * the 146 state-dict keys and shapes of the amazon-config HRqVae, in order."""
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

gin = types.ModuleType("gin")
gin.constants_from_enum = lambda c: c
gin.configurable = lambda f=None, **k: f if f is not None else (lambda g: g)
sys.modules["gin"] = gin
import modules.loss as _L  # noqa: E402  (reference)

_L.CategoricalReconstuctionLoss = _L.CategoricalReconstructionLoss
_tp = types.ModuleType("data.tags_processed")
_tp.ItemData = type("ItemData", (), {})
_tp.SeqData = type("SeqData", (), {})
_tp.RecDataset = type("RecDataset", (), {})
import data  # noqa: E402  (reference package; its __init__ is empty)

sys.modules["data.tags_processed"] = _tp
data.tags_processed = _tp
from data.schemas import SeqBatch  # noqa: E402  (reference)
from modules.h_rqvae import HRqVae  # noqa: E402  (reference)
from modules.quantize import QuantizeForwardMode  # noqa: E402  (reference)
from modules.tokenizer.h_semids import HSemanticIdTokenizer  # noqa: E402  (reference)

from oracle import torch_oracle as O  # noqa: E402

CLASSES = [38, 168, 348]


class Corpus(torch.utils.data.Dataset):
    """item i as the SeqBatch record the reference's DataLoader collates (data/schemas.py:7-13)"""

    def __init__(self, x):
        self.x = x

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        z = torch.zeros((), dtype=torch.long)
        return SeqBatch(user_ids=z, ids=torch.tensor(i), ids_fut=z, x=self.x[i], x_fut=self.x[i], seq_mask=torch.tensor(True))


def make_tok(mode):
    tok = HSemanticIdTokenizer(768, 32, [512, 256, 128], 256, n_layers=3, n_cat_feats=0, hrqvae_codebook_normalize=True,
                               tag_class_counts=CLASSES, use_concatenated_ids=mode == "concat", use_interleaved_ids=mode == "inter")
    cfg = O.Cfg(tag_class_counts=CLASSES)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    sd = tok.hrq_vae.state_dict()
    for k, v in P.items():
        sd[k] = v.clone()
    tok.hrq_vae.load_state_dict(sd)
    return tok, cfg


def main():
    fx = {}
    N = 1100  # three DataLoader batches (512, 512, 76)
    for mode in ("plain", "concat", "inter"):
        tok, cfg = make_tok(mode)
        x, _, _ = O.formula_batch(cfg, N, seed=31, tagged=False)
        with redirect_stdout(io.StringIO()):
            ids = tok.precompute_corpus_ids(Corpus(x))
        fx[f"corpus_ids_{mode}"] = ids.numpy().astype(np.int32)
        print(mode, tuple(ids.shape), "sem_ids_dim", tok.sem_ids_dim, "distinct tuples", len(torch.unique(ids, dim=0)))
        if mode != "plain":
            continue
        corpus = ids.numpy()
        rng = np.random.default_rng(0)
        for width in (1, 2, 3):
            q = corpus[rng.integers(0, N, size=70)][:, :width].copy()
            q[::3, -1] = (q[::3, -1] + 1 + rng.integers(0, 5, size=q[::3].shape[0])) % 256  # every third query perturbed
            with redirect_stdout(io.StringIO()):
                hit = tok.exists_prefix(torch.from_numpy(q))
            fx[f"prefix_q_w{width}"], fx[f"prefix_hit_w{width}"] = q.astype(np.int32), hit.numpy()
            print("  width", width, "hits", int(hit.sum()), "of", q.shape[0], "(rows >= 64 are never examined)")
        # a 3-D prefix [beams, candidates, width] as constrained decoding issues it (modules/model.py of the reference)
        q3 = corpus[rng.integers(0, N, size=40 * 6)][:, :2].reshape(40, 6, 2).copy()
        q3[:, ::2, -1] = (q3[:, ::2, -1] + 3) % 256
        with redirect_stdout(io.StringIO()):
            hit3 = tok.exists_prefix(torch.from_numpy(q3))
        fx["prefix_q_3d"], fx["prefix_hit_3d"] = q3.astype(np.int32), hit3.numpy()
        # a prefix WIDER than the cache is truncated (h_semids.py:207-211)
        q5 = np.concatenate([corpus[:32], np.zeros((32, 2), corpus.dtype)], 1)
        with redirect_stdout(io.StringIO()):
            hit5 = tok.exists_prefix(torch.from_numpy(q5))
        fx["prefix_q_wide"], fx["prefix_hit_wide"] = q5.astype(np.int32), hit5.numpy()
    fx["desc"] = json.dumps(dict(N=N, batch_seed=31, param_seed=100, classes=CLASSES, torch=torch.__version__))
    np.savez_compressed(os.path.join(HERE, "tokenizer_corpus.npz"), **fx)

    # the state-dict contract: keys, shapes, dtypes of the amazon-config model, in order
    m = HRqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, codebook_kmeans_init=False,
               codebook_normalize=True, codebook_mode=QuantizeForwardMode.ROTATION_TRICK, n_layers=3, n_cat_features=0,
               tag_class_counts=CLASSES, tag_embed_dim=768, use_focal_loss=True, dropout_rate=0.4)
    sd = m.state_dict()
    contract = {"config": "configs/h_rqvae_amazon.gin shapes", "n_parameters": int(sum(p.numel() for p in m.parameters())),
                "entries": [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]}
    with open(os.path.join(HERE, "state_dict_contract.json"), "w") as f:
        json.dump(contract, f, indent=0)
    print("state dict:", len(sd), "entries,", contract["n_parameters"], "parameters")


if __name__ == "__main__":
    main()
