"""CPU: pin the fixed-order C oracle (oracle/exact.c) to the reference's golden outputs.
ids must be bit-exact; floats within 1e-5 relative (the C order differs from MKL's, so not bitwise)."""
import numpy as np
import pytest

from oracle import exact, torch_oracle as O
from tests import helpers as H

CASES = [n for n in H.case_names("case") if "gumbel" not in n and "simvq" not in n]


@pytest.mark.parametrize("name", CASES)
def test_exact_oracle_matches_reference(name):
    fx, desc = H.load(name)
    cfg, P, x, _, _ = H.inputs_of(desc)
    y = exact.mlp(x.numpy(), [w.numpy() for w in O.enc_weights(P, cfg)])
    cbs = [P[f"layers.{i}.embedding.weight"].numpy() for i in range(cfg.n_layers)]
    r = exact.rq_forward(y, cbs, cfg.codebook_normalize, cfg.codebook_normalize, cfg.codebook_mode,
                         desc["training"], cfg.commitment_weight)
    safe = H.safe_rows(fx)  # a decision whose top-2 gap is below fp32 noise may legitimately flip: reported, not compared
    assert np.array_equal(r["ids"][safe], fx["sem_ids"].astype(np.int64)[safe])
    H.report_flips(name, r["ids"], fx)
    assert H.rel_err(r["loss"][safe], fx["rqvae_loss"][safe]) <= 1e-5
    if "z" in fx:
        L = cfg.n_layers
        assert H.rel_err(r["z"], fx["z"]) <= 1e-5
        emb = r["emb_cat"].reshape(-1, L, cfg.embed_dim).transpose(0, 2, 1)
        res = r["res_cat"].reshape(-1, L, cfg.embed_dim).transpose(0, 2, 1)
        assert H.rel_err(emb, fx["embeddings"]) <= 1e-5
        assert H.rel_err(res, fx["residuals"]) <= 1e-5
        assert H.rel_err(np.linalg.norm(emb, axis=1), fx["embs_norm"]) <= 1e-5


@pytest.mark.parametrize("name", H.case_names("quantize"))
def test_cosine_ranking_matches_the_reference_quantize_level(name):
    """QuantizeDistance.COSINE (reference modules/quantize.py:115-119) on ONE level: ORDER-GEN with the cosine ranking and the torch
    restatement against the reference's own outputs."""
    import torch
    fx, d = H.load(name)
    x, E, g_out, g_loss = H.quantize_inputs(d)
    assert (fx["margins"] > 1e-6).all()
    r = exact.rq_forward(x, [E], False, d["normalize"], d["mode"], d["training"], d["beta"], cosine=True)
    assert np.array_equal(r["ids"][:, 0], fx["ids"].astype(np.int64))
    assert H.rel_err(r["emb_cat"], fx["embeddings"]) <= 1e-5
    assert H.rel_err(r["loss"], fx["loss"]) <= 1e-5
    xt = torch.from_numpy(x).requires_grad_(d["training"])
    Et = torch.from_numpy(E).requires_grad_(d["training"])
    cb = torch.nn.functional.normalize(Et, dim=-1, eps=1e-12) if d["normalize"] else Et
    out, ids, loss, _ = O.quantize_level(xt, cb, d["mode"], d["beta"], d["training"], 0.2, None, cosine=True)
    assert np.array_equal(ids.numpy(), fx["ids"].astype(np.int64))
    assert H.rel_err(out.detach().numpy(), fx["embeddings"]) <= 1e-5
    if d["training"]:
        ((out * torch.from_numpy(g_out)).sum() + (loss * torch.from_numpy(g_loss)).sum()).backward()
        assert H.close(xt.grad.numpy(), fx["grad_x"], 2e-5, 1e-7)
        assert H.close(Et.grad.numpy(), fx["grad_E"], 2e-5, 1e-7)


def test_own_exp_is_accurate():
    xs = np.linspace(-80, 80, 4001).astype(np.float32)
    got = np.array([exact.lib().orc_exp(float(v)) for v in xs], dtype=np.float64)
    ref = np.exp(xs.astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 3e-7


def test_first_minimum_wins_on_exact_ties():
    y = np.zeros((4, 32), np.float32)
    y[:, 0] = 1.0
    E = np.zeros((16, 32), np.float32)
    E[:, 1] = 1.0  # all 16 codes identical => every distance ties => id 0 (torch.min semantics on CPU)
    r = exact.rq_forward(y, [E, E.copy()], False, False, O.STE, True, 0.25)
    assert (r["ids"] == 0).all()
