"""CPU: checkpoint interchange with the reference (reference train_hidvae.py:621-628 reads, :1161-1171 writes).

The "optimizer" entry is a genuine torch.optim.AdamW.state_dict(); the one pickled object of the file (the QuantizeForwardMode enum
in model_config) is named by the reference's module path, so a file written here loads in an environment WITHOUT hidvae_amd and a
file written by the reference loads here."""
import math
import os
import subprocess
import sys

import pytest
import torch

import hidvae_amd  # noqa: F401
from hidvae_amd.optim import HidvaeAdamW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(input_dim=40, embed_dim=32, hidden_dims=[64, 48], codebook_size=16, codebook_kmeans_init=False, codebook_normalize=True,
           n_layers=2, n_cat_features=0, tag_class_counts=[5, 7], tag_embed_dim=24, use_focal_loss=True,
           focal_loss_params={"gamma_0": 2.7, "alpha_0": 0.24}, dropout_rate=0.4)


def mirror_model():
    from hidvae_amd.modules.h_rqvae import HRqVae
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    torch.manual_seed(3)
    return HRqVae(codebook_mode=QuantizeForwardMode.ROTATION_TRICK, **CFG)


def groups_of(m, lr=2.8e-4, wd=0.015, pwd=0.02):
    """reference train_hidvae.py:533-563"""
    g = [{"params": list(m.encoder.parameters()) + list(m.decoder.parameters()), "lr": lr, "weight_decay": wd},
         {"params": [p for layer in m.layers for p in layer.parameters()], "lr": lr, "weight_decay": wd}]
    for i in range(m.n_layers):
        g.append({"params": list(m.tag_predictors[i].parameters()), "lr": lr * (1 + 0.1 * i), "weight_decay": pwd / (1 + 0.2 * i)})
        g.append({"params": list(m.tag_projectors[i].parameters()), "lr": lr * (1 + 0.1 * i), "weight_decay": pwd / (1 + 0.2 * i)})
    return g


def test_optimizer_state_is_a_torch_adamw_state_dict_both_ways():
    m = mirror_model()
    opt = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6)).prepare()
    g = torch.Generator().manual_seed(0)
    opt._m.copy_(torch.randn(opt._m.shape, generator=g) * 1e-3)
    opt._v.copy_(torch.rand(opt._v.shape, generator=g) * 1e-6)
    opt.step_dev[0] = 7
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["param_groups"]) == 2 + 2 * m.n_layers
    n_params = sum(len(grp["params"]) for grp in sd["param_groups"])
    assert sorted(sd["state"]) == list(range(n_params))
    # -> torch's own optimizer, built with the reference's groups, takes it
    m2 = mirror_model()
    ref = torch.optim.AdamW(groups_of(m2))
    ref.load_state_dict(sd)
    flat = [p for grp in ref.param_groups for p in grp["params"]]
    off = 0
    for p in flat:
        st = ref.state[p]
        n = p.numel()
        assert float(st["step"]) == 7.0
        assert torch.equal(st["exp_avg"].reshape(-1), opt._m[off:off + n]) and torch.equal(st["exp_avg_sq"].reshape(-1), opt._v[off:off + n])
        off += n
    want_lr = 1e-6 + (2.8e-4 - 1e-6) * (1 + math.cos(math.pi * 7 / 1000)) / 2
    assert abs(ref.param_groups[0]["lr"] - want_lr) < 1e-12 and ref.param_groups[0]["initial_lr"] == 2.8e-4
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(ref, T_max=1000, eta_min=1e-6, last_epoch=6)  # the reference's resume (:638)
    for p in flat:
        p.grad = torch.ones_like(p) * 1e-3
    ref.step()
    sched.step()
    # <- and a state written by torch's optimizer loads into the fused one
    sd2 = ref.state_dict()
    m3 = mirror_model()
    opt3 = HidvaeAdamW(groups_of(m3), cosine=(1000, 1e-6)).prepare()
    assert opt3.load_state_dict(sd2) is True
    assert opt3.step_dev.tolist() == [8, 0]
    off = 0
    for p in flat:
        n = p.numel()
        assert torch.equal(ref.state[p]["exp_avg"].reshape(-1), opt3._m[off:off + n])
        assert torch.equal(ref.state[p]["exp_avg_sq"].reshape(-1), opt3._v[off:off + n])
        off += n
    assert opt3.param_groups[2]["lr"] == 2.8e-4 and abs(opt3.param_groups[4]["lr"] - 2.8e-4 * 1.1) < 1e-15  # base rates, not scheduled ones
    # the round-1 private layout is still read
    opt4 = HidvaeAdamW(groups_of(mirror_model()), cosine=(1000, 1e-6)).prepare()
    assert opt4.load_state_dict(opt.flat_state()) is True and torch.equal(opt4._m, opt._m) and int(opt4.step_dev[0]) == 7


def test_resume_without_optimizer_state_restarts_bias_correction_but_not_the_schedule():
    opt = HidvaeAdamW(groups_of(mirror_model()), cosine=(1000, 1e-6), start_step=500).prepare()
    assert opt.load_state_dict({"state": {}, "param_groups": torch.optim.AdamW(groups_of(mirror_model())).state_dict()["param_groups"]}) is False
    assert opt.step_dev.tolist() == [0, 500]
    assert abs(opt.current_lr() - (1e-6 + (2.8e-4 - 1e-6) * 0.5)) < 1e-9  # cosine at half of T_max
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"params": [0]}]})


def test_flat_state_round_trips_into_an_optimizer_with_a_first_bucket():
    """the flat moment layout leads with the data-parallel first bucket, so a flat state must be scattered parameter by
    parameter (positional copies put Adam moments on the wrong parameters; sizes match, nothing raised)"""
    m = mirror_model()
    opt = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6)).prepare()
    opt._m.copy_(torch.arange(opt._m.numel(), dtype=torch.float32))
    opt._v.copy_(torch.arange(opt._v.numel(), dtype=torch.float32) * 2)
    opt.step_dev[0] = 9
    per_param = {id(p): opt._m[o:o + n].clone() for p, (o, n) in ((p, opt._slots()[id(p)]) for p in opt._params)}
    # same model object, another optimizer whose flat order leads with the decoder's last two layers
    first = list(m.decoder.parameters())[-2:]
    opt2 = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6), first_bucket=first).prepare()
    assert opt2.n_first == 2 and opt2._params[0] is first[0] and opt2._params[0] is not opt._params[0]
    fs = opt.flat_state()
    assert opt2.load_state_dict(fs) is True
    for p in opt2._params:
        o, n = opt2._slots()[id(p)]
        assert torch.equal(opt2._m[o:o + n], per_param[id(p)]), "moment landed on another parameter"
    # and back: a state written WITH a first bucket loads into a plain optimizer
    opt3 = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6)).prepare()
    assert opt3.load_state_dict(opt2.flat_state()) is True and torch.equal(opt3._m, opt._m) and torch.equal(opt3._v, opt._v)
    # a round-1 file (no order recorded) is in group order
    legacy = {k: v for k, v in fs.items() if k not in ("order", "numels", "schedule_offset")}
    opt4 = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6), first_bucket=first).prepare()
    assert opt4.load_state_dict(legacy) is True
    for p in opt4._params:
        o, n = opt4._slots()[id(p)]
        assert torch.equal(opt4._m[o:o + n], per_param[id(p)])
    bad = dict(fs, numels=[1] + fs["numels"][1:])
    with pytest.raises(ValueError):
        opt4.load_state_dict(bad)


def test_resuming_twice_keeps_the_schedule_position():
    """state_dict() dropped the schedule offset of a run resumed without optimizer state"""
    opt = HidvaeAdamW(groups_of(mirror_model()), cosine=(1000, 1e-6)).prepare()
    opt.restart_without_state(500)
    opt.step_dev[0] = 3  # three optimizer steps after the restart
    opt._m.fill_(0.5)
    sd = opt.state_dict()
    opt2 = HidvaeAdamW(groups_of(mirror_model()), cosine=(1000, 1e-6)).prepare()
    assert opt2.load_state_dict(sd) is True
    assert opt2.step_dev.tolist() == [3, 500] and abs(opt2.current_lr() - opt.current_lr()) < 1e-12
    torch.optim.AdamW(groups_of(mirror_model())).load_state_dict(sd)  # the extra key does not disturb torch's own loader


REFERENCE_SIDE = r'''
import sys, types, json
sys.path.insert(0, "/root/reference")
g = types.ModuleType("gin"); g.configurable = lambda f=None, **k: f if f is not None else (lambda h: h); g.constants_from_enum = lambda c: c
sys.modules["gin"] = g
import torch
import modules.loss as L
L.CategoricalReconstuctionLoss = L.CategoricalReconstructionLoss
from modules.h_rqvae import HRqVae
from modules.quantize import QuantizeForwardMode
assert "hidvae_amd" not in sys.modules
src, dst = sys.argv[1], sys.argv[2]
state = torch.load(src, map_location="cpu", weights_only=False)           # train_hidvae.py:624
cfg = state["model_config"]
assert cfg["codebook_mode"] is QuantizeForwardMode.ROTATION_TRICK, cfg["codebook_mode"]
model = HRqVae(**cfg)
model.load_state_dict(state["model"])                                      # strict: the 146-key contract, both directions
lr, wd, pwd = 2.8e-4, 0.015, 0.02
groups = [{"params": list(model.encoder.parameters()) + list(model.decoder.parameters()), "lr": lr, "weight_decay": wd},
          {"params": [p for layer in model.layers for p in layer.parameters()], "lr": lr, "weight_decay": wd}]
for i in range(cfg["n_layers"]):
    groups.append({"params": model.tag_predictors[i].parameters(), "lr": lr * (1 + i * 0.1), "weight_decay": pwd / (1 + i * 0.2)})
    groups.append({"params": model.tag_projectors[i].parameters(), "lr": lr * (1 + i * 0.1), "weight_decay": pwd / (1 + i * 0.2)})
opt = torch.optim.AdamW(groups)
opt.load_state_dict(state["optimizer"])                                    # train_hidvae.py:625
start_iter = state["iter"] + 1
sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=1000, eta_min=1e-6, last_epoch=start_iter - 1)   # :638
for p in model.parameters():
    p.grad = torch.full_like(p, 1e-3)
opt.step(); sched.step()
out = {"iter": start_iter, "model": model.state_dict(), "model_config": model.config, "optimizer": opt.state_dict(),
       "accuracy": 0.7, "rqvae_loss": 0.1, "sem_id_repetition_rate": 0.01}  # :1161-1169 (config pickles the module too, SURVEY Q11)
torch.save(out, dst)
print("reference-side ok", len(state["model"]), float(list(opt.state.values())[0]["step"]))
'''


@pytest.mark.skipif(not os.path.exists("/root/reference/modules/h_rqvae.py"), reason="the reference tree only exists in the build container")
def test_checkpoint_round_trip_through_the_reference(tmp_path):
    from hidvae_amd.checkpoint import load_checkpoint, save_checkpoint
    m = mirror_model()
    opt = HidvaeAdamW(groups_of(m), cosine=(1000, 1e-6)).prepare()
    g = torch.Generator().manual_seed(1)
    opt._m.copy_(torch.randn(opt._m.shape, generator=g) * 1e-3)
    opt._v.copy_(torch.rand(opt._v.shape, generator=g) * 1e-6)
    opt.step_dev[0] = 12
    ours, theirs = str(tmp_path / "ours.pt"), str(tmp_path / "theirs.pt")
    save_checkpoint({"iter": 12, "model": m.state_dict(), "model_config": m.config, "optimizer": opt.state_dict(), "accuracy": 0.65,
                     "rqvae_loss": 0.2, "sem_id_repetition_rate": 0.02}, ours)
    from hidvae_amd.modules.quantize import QuantizeForwardMode
    assert QuantizeForwardMode.__module__ == "hidvae_amd.modules.quantize" and "modules.quantize" not in sys.modules  # aliases were temporary
    env = {**os.environ, "TORCHDYNAMO_DISABLE": "1"}
    out = subprocess.run([sys.executable, "-c", REFERENCE_SIDE, ours, theirs], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert out.returncode == 0 and "reference-side ok" in out.stdout, out.stderr[-3000:]
    # the file the REFERENCE wrote (its own torch optimizer, its own pickled enum and module) loads here
    state = load_checkpoint(theirs, map_location="cpu")
    assert state["iter"] == 13 and state["model_config"]["codebook_mode"] is QuantizeForwardMode.ROTATION_TRICK
    assert "self" not in state["model_config"]
    m2 = mirror_model()
    m2.load_state_dict(state["model"])
    assert list(m2.state_dict()) == list(state["model"])
    opt2 = HidvaeAdamW(groups_of(m2), cosine=(1000, 1e-6)).prepare()
    assert opt2.load_state_dict(state["optimizer"]) is True
    assert int(opt2.step_dev[0]) == 13
    # one torch AdamW step on constant gradients from our moments: m' = 0.9 m + 0.1 g, v' = 0.999 v + 0.001 g^2
    assert torch.allclose(opt2._m, 0.9 * opt._m + 0.1 * 1e-3, rtol=1e-5, atol=1e-9)
    assert torch.allclose(opt2._v, 0.999 * opt._v + 0.001 * 1e-6, rtol=1e-5, atol=1e-12)

