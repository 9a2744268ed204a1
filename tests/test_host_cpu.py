"""CPU: host-side logic that needs no kernel -- the gin subset parser, the rare-tag remap, package plumbing."""
import os

import pytest
import torch

import hidvae_amd  # noqa: F401
from hidvae_amd import gin_compat as gin

CONFIG = '''
import data.tags_processed
import modules.quantize

# comment line
train.iterations=400000
train.learning_rate=0.00028
train.batch_size=128
train.vae_hidden_dims=[512, 256, 128]
train.vae_codebook_normalize=True
train.dataset=%data.tags_processed.RecDataset.AMAZON
train.save_dir_root="out/hrqvae/amazon/"
train.vae_codebook_mode=%modules.quantize.QuantizeForwardMode.ROTATION_TRICK
train.tag_class_counts=[38, 168, 348]   # trailing comment
train.lr_scheduler_type='cosine'
train.lr_scheduler_eta_min=7e-8
'''


def test_gin_subset_parser_binds_train_arguments():
    from hidvae_amd.data.items import RecDataset
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    gin.clear_config()
    gin.parse_config(CONFIG, import_aliases={"data.tags_processed": "hidvae_amd.data.items", "modules.quantize": "hidvae_amd.modules.quantize"})
    b = gin.bindings("train")
    assert b["iterations"] == 400000 and b["learning_rate"] == 0.00028 and b["batch_size"] == 128
    assert b["vae_hidden_dims"] == [512, 256, 128] and b["vae_codebook_normalize"] is True
    assert b["vae_codebook_mode"] is QuantizeForwardMode.ROTATION_TRICK and b["dataset"] is RecDataset.AMAZON
    assert b["tag_class_counts"] == [38, 168, 348] and b["lr_scheduler_type"] == "cosine" and b["lr_scheduler_eta_min"] == 7e-8
    assert b["save_dir_root"] == "out/hrqvae/amazon/"

    @gin.configurable
    def train(iterations=1, batch_size=2, **kw):
        return iterations, batch_size, kw

    with pytest.raises(TypeError):
        train()  # unknown bound names are rejected, like gin
    gin.clear_config()


@pytest.mark.skipif(not os.path.exists("/root/reference/configs"), reason="reference configs only exist in the build container")
@pytest.mark.parametrize("cfg", ["h_rqvae_amazon.gin", "h_rqvae_kuairand.gin"])
def test_reference_gin_files_parse(cfg):
    """Every binding of the two tokenizer configs is accepted and names an argument of train()."""
    import inspect
    from hidvae_amd.train_hidvae import train
    gin.clear_config()
    gin.parse_config_file(os.path.join("/root/reference/configs", cfg),
                          import_aliases={"data.tags_processed": "hidvae_amd.data.items", "modules.quantize": "hidvae_amd.modules.quantize"})
    b = gin.bindings("train")
    params = set(inspect.signature(train.__wrapped__).parameters)
    assert b and set(b) <= params, set(b) - params
    assert b["vae_codebook_mode"].name == "ROTATION_TRICK" and b["vae_n_layers"] == 3 and b["vae_embed_dim"] == 32
    gin.clear_config()


def test_rare_tag_remap_matches_literal_restatement():
    from hidvae_amd.train_hidvae import remap_rare_tags
    g = torch.Generator().manual_seed(0)
    ti = torch.stack([torch.randint(0, c, (4000,), generator=g) for c in (12, 40)], dim=1)
    ti[torch.rand(ti.shape, generator=g) < 0.1] = -1
    ti[:, 1][ti[:, 1] > 30] = -1  # classes 31..39 never occur
    ev = ti[:500].clone()
    want_tr, want_ev = ti.clone(), ev.clone()
    new_counts = []
    for i, C in enumerate((12, 40)):  # reference train_hidvae.py:366-455, written out literally
        col = ti[:, i]
        counts = torch.zeros(C, dtype=torch.long)
        u, c = torch.unique(col[col >= 0], return_counts=True)
        counts[u] = c
        rare = torch.nonzero((counts > 0) & (counts < 110)).squeeze(-1)
        new_counts.append(int(((counts >= 110) | (counts == 0)).sum()) + 1)
        keep = torch.ones(C, dtype=torch.bool)
        keep[rare] = False
        mapping = torch.arange(C)
        mapping[keep] = (torch.cumsum(keep, 0) - 1)[keep]
        mapping[rare] = new_counts[-1] - 1
        for w in (want_tr, want_ev):
            ok = w[:, i] >= 0
            w[ok, i] = mapping[w[ok, i]]
    tr, evc = ti.clone(), ev.clone()
    got_counts, rare_ids, _ = remap_rare_tags(tr, evc, [12, 40], 2, 110)
    assert got_counts == new_counts
    assert torch.equal(tr, want_tr) and torch.equal(evc, want_ev)


def test_install_dropin_aliases_reference_module_names():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import hidvae_amd; hidvae_amd.install_dropin(); "
            "from modules.h_rqvae import HRqVae; from modules.quantize import QuantizeForwardMode, Quantize; "
            "from data.schemas import HRqVaeComputedLosses, TaggedSeqBatch; from modules.tokenizer.h_semids import HSemanticIdTokenizer; "
            "from init.kmeans import kmeans_init_; from modules.rqvae import RqVae; "
            "from ops.triton.jagged import padded_to_jagged_tensor, jagged_to_flattened_tensor; "
            "assert RqVae.__module__ == 'hidvae_amd.modules.rqvae' and padded_to_jagged_tensor.__module__ == 'hidvae_amd.ops_hip.jagged'; "
            "assert HRqVae.__module__ == 'hidvae_amd.modules.h_rqvae' and QuantizeForwardMode.ROTATION_TRICK.value == 3; "
            "assert HRqVaeComputedLosses._fields[0] == 'loss' and len(HRqVaeComputedLosses._fields) == 12; print('ok')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.skipif(not os.path.exists("/root/reference/modules/utils.py"), reason="the reference tree only exists in the build container")
def test_install_dropin_keeps_unmirrored_reference_modules_importable():
    """install_dropin() replaces LEAF modules only: the reference's own packages stay in place, so the import block of its
    train_hidvae.py (`from modules.utils import parse_config`, line 9) and the stage-2 modules still resolve to its files, while the
    mirrored names resolve to the HIP implementation.  (`gin` is not installed here: a decorator stub stands in, as in
    tests/golden/make_golden.py.)"""
    import subprocess
    import sys
    code = ("import sys, types; sys.path.insert(0, %r); sys.path.insert(0, '/root/reference'); "
            "g = types.ModuleType('gin'); g.configurable = lambda f=None, **k: f if f is not None else (lambda h: h); "
            "g.constants_from_enum = lambda c: c; g.parse_config_file = lambda *a, **k: None; sys.modules['gin'] = g; "
            "import hidvae_amd; hidvae_amd.install_dropin(); "
            "import modules, modules.utils, modules.transformer.attention, modules.embedding.id_embedder; "
            "from modules.utils import parse_config; "
            "assert modules.__file__.startswith('/root/reference') and modules.utils.__file__.startswith('/root/reference'); "
            "from modules.h_rqvae import HRqVae; from modules.quantize import QuantizeForwardMode; "
            "from modules.tokenizer.h_semids import HSemanticIdTokenizer; from data.utils import batch_to, cycle, next_batch; "
            "from distributions.gumbel import TemperatureScheduler, gumbel_softmax_sample; from modules.loss import ReconstructionLoss; "
            "import modules.h_rqvae; assert modules.h_rqvae is sys.modules['modules.h_rqvae']; "
            "assert HRqVae.__module__ == 'hidvae_amd.modules.h_rqvae' and HSemanticIdTokenizer.__module__.startswith('hidvae_amd.'); "
            "assert gumbel_softmax_sample.__module__ == 'hidvae_amd.distributions.gumbel'; "
            "assert ReconstructionLoss.__module__ == 'hidvae_amd.modules.loss'; print('ok')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_reference_processed_dataset_file_loads_without_torch_geometric(tmp_path):
    """The reference writes its dataset with torch_geometric's InMemoryDataset.save: torch.save((HeteroData.to_dict(), slices,
    <class HeteroData>), legacy serialization) (data/tags_amazon.py:392-431).  The loader must read it with torch_geometric
    absent: a look-alike class is registered only while the file is WRITTEN, then removed."""
    import sys
    import types
    import numpy as np
    import torch
    from hidvae_amd.data.items import load_reference_processed
    assert "torch_geometric" not in sys.modules
    pkg, sub, leaf = types.ModuleType("torch_geometric"), types.ModuleType("torch_geometric.data"), types.ModuleType("torch_geometric.data.hetero_data")
    HeteroData = type("HeteroData", (), {"__module__": "torch_geometric.data.hetero_data"})
    leaf.HeteroData = HeteroData
    sys.modules.update({"torch_geometric": pkg, "torch_geometric.data": sub, "torch_geometric.data.hetero_data": leaf})
    g = torch.Generator().manual_seed(0)
    item = {"x": torch.randn(50, 768, generator=g), "text": np.array(["t%d" % i for i in range(50)]),
            "tags_emb": torch.randn(50, 3, 768, generator=g), "tags": np.array([["a", "b", "c"]] * 50),
            "tags_indices": torch.randint(-1, 30, (50, 3), generator=g), "is_train": torch.rand(50, generator=g) > 0.05}
    payload = ({"_global_store": {}, "item": item, "user": {"x": torch.zeros(4, 2)}, ("user", "rated", "item"): {"edge_index": torch.zeros(2, 3)}},
               None, HeteroData)
    try:
        for legacy in (False, True):  # the reference forces the legacy (non-zipfile) serialization
            torch.save(payload, str(tmp_path / f"d{int(legacy)}.pt"), _use_new_zipfile_serialization=not legacy)
    finally:
        for k in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.hetero_data"):
            sys.modules.pop(k, None)
    for legacy in (False, True):
        got = load_reference_processed(str(tmp_path / f"d{int(legacy)}.pt"))
        assert set(got) == {"x", "tags_emb", "tags_indices", "is_train"}
        for k in got:
            assert torch.equal(got[k], item[k]), k
    plain = {"x": item["x"], "is_train": item["is_train"]}
    torch.save(plain, str(tmp_path / "plain.pt"))
    assert set(load_reference_processed(str(tmp_path / "plain.pt"))) == {"x", "is_train"}


def test_state_dict_contract_matches_the_reference():
    """Checkpoint interchange rests on the state-dict keys: the amazon-config mirror must expose exactly the reference's
    146 entries -- names, order, shapes, dtypes (tests/golden/state_dict_contract.json, dumped from the reference's own HRqVae by
    tests/golden/make_golden_tokenizer.py)."""
    import json
    from hidvae_amd.modules.h_rqvae import HRqVae
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "state_dict_contract.json")))
    m = HRqVae(input_dim=768, embed_dim=32, hidden_dims=[512, 256, 128], codebook_size=256, codebook_kmeans_init=False,
               codebook_normalize=True, codebook_mode=QuantizeForwardMode.ROTATION_TRICK, n_layers=3, n_cat_features=0,
               tag_class_counts=[38, 168, 348], tag_embed_dim=768, use_focal_loss=True, dropout_rate=0.4)
    got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    assert len(got) == len(want["entries"]) == 146
    assert got == want["entries"]
    assert sum(p.numel() for p in m.parameters()) == want["n_parameters"] == 7244236


def test_flat_grad_buffer_first_bucket_is_self_checking():
    """a first bucket that the first half of a split backward did not complete must raise, not be zero-filled; and a write
    into the first bucket during the second half (while its all-reduce is in flight) must raise as well"""
    import pytest
    import torch
    from hidvae_amd.ops import grad_sink
    from hidvae_amd.parallel import FlatGradBuffer
    ps = [torch.nn.Parameter(torch.zeros(4)) for _ in range(3)]
    buf = FlatGradBuffer(ps)
    # (a) slot 1 of the first bucket not produced
    buf.zero()
    dst, acc = grad_sink(ps[0])
    dst.fill_(1.0)
    with pytest.raises(RuntimeError, match="were not produced"):
        buf.seal(upto=2)
    # (b) a kernel-style late write into the first bucket
    buf.zero()
    for p in ps[:2]:
        grad_sink(p)[0].fill_(2.0)
    buf.seal(upto=2)
    grad_sink(ps[1])
    with pytest.raises(RuntimeError, match="second half"):
        buf.check_first_bucket_untouched()
    # (c) an autograd-style late gradient for a first-bucket parameter
    buf.zero()
    for p in ps[:2]:
        grad_sink(p)[0].fill_(2.0)
    buf.seal(upto=2)
    ps[0].grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="second half"):
        buf.seal()
    # (d) the good path: the bucket is left alone, the rest arrives, everything ends up as views of the flat buffer
    buf.zero()
    for p in ps[:2]:
        grad_sink(p)[0].fill_(3.0)
    buf.seal(upto=2)
    ps[2].grad = torch.full((4,), 5.0)
    buf.check_first_bucket_untouched()
    buf.seal()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ps, buf.views))
    assert buf.flat.tolist() == [3.0] * 8 + [5.0] * 4


def test_optimizer_tensor_ranges_for_an_early_update():
    """HidvaeAdamW.tensor_ranges_of: where a set of parameters sits in the optimizer's tensor tables, as contiguous ranges -- what
    GraphedTrainStep hands step_early() for a tag level's heads (the update itself is a HIP launch: GPU tests)"""
    import torch
    from hidvae_amd.optim import HidvaeAdamW
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (3, 5, 2, 7, 4, 1)]
    frozen = torch.nn.Parameter(torch.zeros(2), requires_grad=False)
    opt = HidvaeAdamW([{"params": ps[:2]}, {"params": ps[2:4] + [frozen]}, {"params": ps[4:]}], lr=1e-3)
    assert opt.tensor_ranges_of(ps[2:4]) == [(2, 4)]
    assert opt.tensor_ranges_of([ps[0], ps[1], ps[4], ps[5]]) == [(0, 2), (4, 6)]
    assert opt.tensor_ranges_of([ps[3], frozen]) == [(3, 4)]           # a frozen parameter is not an optimized tensor: ignored
    assert opt.tensor_ranges_of([torch.nn.Parameter(torch.zeros(1))]) is None   # a stranger: no early update for that set
    assert opt.step_early([(0, 2)]) is False                            # nothing prepared (and CPU tensors): it must decline, not raise


def test_random_batches_gathered_into_given_buffers_equal_fresh_batches():
    """RandomBatches.next(out=...) (the training loop hands it GraphedTrainStep.input_buffers()): the same rows as next(), written in place"""
    import types
    from hidvae_amd.data.items import RandomBatches, ResidentItemData
    g = torch.Generator().manual_seed(3)
    data = ResidentItemData(torch.randn(50, 8, generator=g), torch.randn(50, 3, 8, generator=g), torch.randint(0, 9, (50, 3), generator=g))
    a, b = RandomBatches(data, 16, seed=5), RandomBatches(data, 16, seed=5)
    out = types.SimpleNamespace(x=torch.empty(16, 8), tags_emb=torch.empty(16, 3, 8), tags_indices=torch.empty(16, 3, dtype=torch.long))
    px = out.x.data_ptr()
    for _ in range(7):  # crosses a re-permutation (50 items, 16 per batch)
        fresh, given = a.next(), b.next(out=out)
        assert given is out and out.x.data_ptr() == px
        assert torch.equal(fresh.x, out.x) and torch.equal(fresh.tags_emb, out.tags_emb) and torch.equal(fresh.tags_indices, out.tags_indices)
    short = types.SimpleNamespace(x=torch.empty(8, 8))  # buffers of another batch size are left alone
    assert b.next(out=short) is not short
