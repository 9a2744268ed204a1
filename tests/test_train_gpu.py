"""GPU: the training loop pieces against the reference's 3-iteration golden run, and the train() driver end to end."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from tests import helpers as H
from tests.test_model_gpu import build_model, make_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flat", [False, True], ids=["per-tensor-grads", "flat-grad-buffer"])
@pytest.mark.parametrize("name", H.case_names("train"))
def test_training_iterations_match_the_reference_run(name, flat):
    """(3 iterations with 2 accumulated micro-batches, and a 20-iteration loss sequence)
    grad accumulation (several forwards, one backward), 8 AdamW param groups with layer-specific lr / wd, cosine schedule:
    losses per iteration, final codebooks and BatchNorm buffers vs the reference's own run (train_hidvae.py:533-563,698-766)."""
    from hidvae_amd.optim import HidvaeAdamW
    from hidvae_amd.rand import InjectedRand
    fx, desc = H.load(name)
    cfg = H.cfg_of(desc)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    m = build_model(cfg, P).train()
    m.rand = InjectedRand(O.FormulaRand(**desc["rand"]))
    lr, wd, pwd = desc["lr"], desc["wd"], desc["pwd"]
    groups = [{"params": list(m.encoder.parameters()) + list(m.decoder.parameters()), "lr": lr, "weight_decay": wd},
              {"params": [p for layer in m.layers for p in layer.parameters()], "lr": lr, "weight_decay": wd}]
    for i in range(cfg.n_layers):
        groups.append({"params": list(m.tag_predictors[i].parameters()), "lr": lr * (1 + 0.1 * i), "weight_decay": pwd / (1 + 0.2 * i)})
        groups.append({"params": list(m.tag_projectors[i].parameters()), "lr": lr * (1 + 0.1 * i), "weight_decay": pwd / (1 + 0.2 * i)})
    # flat=True: the data-parallel layout (one gradient buffer; kernels write their slots in place, parallel.FlatGradBuffer)
    opt = HidvaeAdamW(groups, cosine=(desc["T_max"], desc["eta_min"]), flat_grads=flat)
    losses = []
    for it in range(desc["iters"]):
        opt.zero_grad()
        total = 0
        for a in range(desc["ga"]):
            x, te, ti = O.formula_batch(cfg, desc["B"], seed=1000 + 37 * (it * desc["ga"] + a), tagged=True)
            total = total + m(make_batch(x, te, ti), gumbel_t=0.2).loss / desc["ga"]
        total.backward()
        losses.append(float(total.detach()))
        opt.step()
    assert H.rel_err(np.array(losses), fx["losses"]) <= 1e-5, (losses, fx["losses"])
    sd = m.state_dict()
    loose = H.zero_grad_keys(cfg)
    for k in fx:
        if k.startswith("param/"):
            assert H.close(sd[k[6:]].cpu().numpy(), fx[k], 1e-5, 0.02 * lr * desc["iters"]), k
        if k.startswith("psample/"):
            # Adam turns 1e-6-relative gradient noise into O(lr) steps wherever |g| ~ eps (see test_model_gpu): the bulk must
            # agree tightly, no element may be further off than a fraction of the distance it could have travelled
            d = np.abs(H.sample(sd[k[8:]]).astype(np.float64) - fx[k])
            assert d.max() <= (2.0 if k[8:] in loose else 0.5) * lr * desc["iters"], k
            if k[8:] not in loose:
                assert np.median(d) <= 0.01 * lr * desc["iters"], k
    for i in range(cfg.n_layers):
        # (the projector weights feeding BatchNorm have themselves moved by Adam-amplified noise after 3 steps)
        # (... and the running mean contains the projector's first bias, whose gradient is mathematically zero: Adam random-walks it)
        assert H.close(sd[f"tag_projectors.{i}.1.running_mean"].cpu().numpy(), fx[f"bn_mean_{i}"], 5e-5, 0.05 * lr * desc["iters"])
        assert H.rel_err(sd[f"tag_projectors.{i}.1.running_var"].cpu().numpy(), fx[f"bn_var_{i}"]) <= 5e-5
        assert int(sd[f"tag_projectors.{i}.1.num_batches_tracked"]) == desc["iters"] * desc["ga"]


@pytest.mark.parametrize("ga", [2, 1])
def test_train_driver_end_to_end(tmp_path, ga):
    """train(): synthetic resident items, k-means warm-up, focal + rare-tag remap, eval with TTA and id diversity, series dump.
    (ga = 1 is a regression case: the k-means warm-up is a forward without a backward; autograd nodes that held their own outputs
    kept that pass's graph -- and its AccumulateGrad nodes on the default stream -- alive, which broke the HIP-graph capture of
    the step a few iterations later.)"""
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    from hidvae_amd.train_hidvae import train
    model, series = train(
        iterations=30, batch_size=128, learning_rate=2.8e-4, weight_decay=0.015, dataset="synthetic:3000", save_dir_root=str(tmp_path) + "/",
        use_kmeans_init=True, do_eval=True, gradient_accumulate_every=ga, eval_every=30, commitment_weight=0.4, tag_alignment_weight=0.15,
        tag_prediction_weight=0.55, vae_n_cat_feats=0, vae_input_dim=768, vae_embed_dim=32, vae_hidden_dims=[512, 256, 128],
        vae_codebook_size=256, vae_codebook_normalize=True, vae_codebook_mode=QuantizeForwardMode.ROTATION_TRICK, vae_n_layers=3,
        tag_class_counts=[38, 168, 348], use_focal_loss=True, focal_loss_gamma_base=2.7, focal_loss_alpha_base=0.24, rare_tag_threshold=5,
        dropout_rate=0.4, lr_scheduler_T_max=400000, lr_scheduler_eta_min=7e-8, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0,
        id_repetition_threshold=0.06, log_every=10)
    assert all(layer.kmeans_initted for layer in model.layers)
    ls = [row[0] for row in series["loss"]]
    assert len(ls) >= 3 and all(np.isfinite(ls)) and ls[-1] < ls[0]
    ev = series["eval"][-1]
    for k in ("eval_total_loss", "eval_tag_pred_accuracy", "rqvae_entropy", "sem_id_repetition_rate", "codebook_usage", "tta_accuracy_by_layer"):
        assert k in ev
    assert 0.0 <= ev["sem_id_repetition_rate"] < 1.0 and len(ev["codebook_usage"]) == 3
    assert os.path.exists(os.path.join(str(tmp_path), "special_tags_files", "rare_tags.pt"))


def test_checkpoint_round_trip(tmp_path):
    """The reference's checkpoint dict (train_hidvae.py:1161-1169) written and read back through load_pretrained."""
    cfg = O.Cfg(tag_class_counts=[38, 168, 348])
    P = O.formula_params(cfg, seed=100, with_tags=True)
    m = build_model(cfg, P)
    path = os.path.join(str(tmp_path), "ck.pt")
    torch.save({"iter": 41, "model": m.state_dict(), "model_config": m.config, "optimizer": {}, "accuracy": 0.7, "rqvae_loss": 0.3,
                "sem_id_repetition_rate": 0.01}, path)
    cfg2 = O.Cfg(tag_class_counts=[7, 30, 97])  # different head widths: the loader rebuilds the predictors (h_rqvae.py:414-430)
    m2 = build_model(cfg2, O.formula_params(cfg2, seed=5, with_tags=True))
    m2.load_pretrained(path)
    assert m2.tag_class_counts == [38, 168, 348]
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k].cpu(), v.cpu()), k
    assert "self" not in m.config and m.config["codebook_size"] == 256


def test_train_driver_graph_replay_equals_eager(tmp_path):
    """train() replays the step from a HIP graph (hidvae_amd/step.py); the untagged step draws nothing at random inside the step,
    so the replayed run must equal the eager run bit for bit: every logged row and every final parameter."""
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    from hidvae_amd.train_hidvae import train
    g = torch.Generator().manual_seed(11)
    x = torch.nn.functional.normalize(torch.randn(1500, 768, generator=g), dim=-1)

    def run(graph, sub):
        return train(iterations=14, batch_size=128, learning_rate=2.8e-4, weight_decay=0.015, dataset={"x": x}, save_dir_root=str(tmp_path / sub) + "/",
                     use_kmeans_init=False, do_eval=False, gradient_accumulate_every=1, commitment_weight=0.4, vae_n_cat_feats=0,
                     vae_input_dim=768, vae_embed_dim=32, vae_hidden_dims=[512, 256, 128], vae_codebook_size=256, vae_codebook_normalize=True,
                     vae_codebook_mode=QuantizeForwardMode.ROTATION_TRICK, vae_n_layers=3, tag_class_counts=[38, 168, 348],
                     lr_scheduler_T_max=1000, lr_scheduler_eta_min=7e-8, log_every=1, seed=3, use_hip_graph=graph)

    m1, s1 = run(True, "g")
    m0, s0 = run(False, "e")
    assert len(s1["loss"]) == len(s0["loss"]) >= 14
    assert s1["loss"] == s0["loss"]
    for (k, a), (_, b) in zip(m1.state_dict().items(), m0.state_dict().items()):
        assert torch.equal(a, b), k
