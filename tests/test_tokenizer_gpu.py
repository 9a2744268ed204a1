"""GPU: HSemanticIdTokenizer (corpus ids, prefix search, sequence tokenisation) against a literal restatement of the
reference's semantics built from the oracle's ids."""
import types

import numpy as np
import pytest
import torch

from oracle import exact, torch_oracle as O

pytestmark = pytest.mark.gpu


def make_tok(mode=None):
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.tokenizer.h_semids import HSemanticIdTokenizer
    kw = dict(use_concatenated_ids=mode == "concat", use_interleaved_ids=mode == "inter")
    cfg = O.Cfg(tag_class_counts=[38, 168, 348])
    tok = HSemanticIdTokenizer(768, 32, [512, 256, 128], 256, n_layers=3, n_cat_feats=0, hrqvae_codebook_normalize=True,
                               tag_class_counts=[38, 168, 348], **kw)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    sd = tok.hrq_vae.state_dict()
    for k, v in P.items():
        sd[k] = v.clone()
    tok.hrq_vae.load_state_dict(sd)
    return tok.cuda(), cfg, P


def oracle_ids(cfg, P, x):
    y = exact.mlp(x.numpy(), [w.numpy() for w in O.enc_weights(P, cfg)])
    return exact.rq_forward(y, [P[f"layers.{i}.embedding.weight"].numpy() for i in range(3)], True, True, 3, False, 0.25)["ids"]


def test_corpus_ids_and_prefix_search():
    tok, cfg, P = make_tok()
    x, _, _ = O.formula_batch(cfg, 700, seed=31, tagged=False)
    ids = tok.precompute_corpus_ids(x)
    want = oracle_ids(cfg, P, x)
    assert ids.shape == (700, 3) and ids.dtype == torch.int64
    assert np.array_equal(ids.cpu().numpy(), want)
    # prefix queries: real prefixes, perturbed ones, every prefix length; compare with the reference's literal expression
    rng = np.random.default_rng(0)
    for width in (1, 2, 3):
        q = want[rng.integers(0, 700, size=70)][:, :width].copy()
        q[::3, -1] = (q[::3, -1] + 1 + rng.integers(0, 5, size=q[::3].shape[0])) % 256
        got = tok.exists_prefix(torch.from_numpy(q).cuda()).cpu().numpy()
        lit = (q[:, None, :] == want[None, :, :width]).all(-1).any(-1)
        covered = (q.shape[0] // 16) * 16  # rows the reference actually examines (h_semids.py:218)
        assert np.array_equal(got[:covered], lit[:covered])
        assert not got[covered:].any()


@pytest.mark.parametrize("mode", [None, "concat", "inter"])
def test_corpus_ids_match_the_reference_tokenizer(mode):
    """precompute_corpus_ids against the REFERENCE's own HSemanticIdTokenizer run on the same items and weights
    (tests/golden/tokenizer_corpus.npz, made by tests/golden/make_golden_tokenizer.py; reference h_semids.py:109-197): semantic ids
    and, in the combined modes, the eval-mode tag predictions placed as the reference places them."""
    from tests import helpers as H
    fx, desc = H.load("tokenizer_corpus")
    tok, cfg, P = make_tok(mode)
    x, _, _ = O.formula_batch(cfg, desc["N"], seed=desc["batch_seed"], tagged=False)
    got = tok.precompute_corpus_ids(x).cpu().numpy()
    want = fx["corpus_ids_" + {None: "plain", "concat": "concat", "inter": "inter"}[mode]].astype(np.int64)
    assert got.shape == want.shape and got.dtype == np.int64
    assert np.array_equal(got, want), f"{int((got != want).any(1).sum())} of {len(want)} corpus rows differ from the reference's"


def test_exists_prefix_matches_the_reference_truth_tables():
    """exists_prefix (sorted-key binary search here, a [queries, corpus, width] broadcast compare in the reference,
    h_semids.py:199-239) against the reference's own answers: every prefix width, a 3-D prefix, a prefix wider than the cache,
    and the trailing rows % 16 rows the reference never examines."""
    from tests import helpers as H
    fx, desc = H.load("tokenizer_corpus")
    tok, cfg, P = make_tok()
    tok.cached_ids = torch.from_numpy(fx["corpus_ids_plain"].astype(np.int64)).cuda()  # the reference's cache itself
    for key in ("w1", "w2", "w3", "3d", "wide"):
        q = torch.from_numpy(fx["prefix_q_" + key].astype(np.int64)).cuda()
        got = tok.exists_prefix(q)
        assert got.dtype == torch.bool and tuple(got.shape) == tuple(fx["prefix_hit_" + key].shape)
        assert np.array_equal(got.cpu().numpy(), fx["prefix_hit_" + key]), key
    assert not fx["prefix_hit_w3"][64:].any() and fx["prefix_hit_w3"][:64].any()  # (the fixture does contain the skipped-rows quirk)


def test_sequence_tokenisation_cached_and_uncached():
    tok, cfg, P = make_tok()
    x, _, _ = O.formula_batch(cfg, 64, seed=32, tagged=False)
    want = oracle_ids(cfg, P, x)
    B, N = 4, 5
    item = torch.arange(B * N).reshape(B, N) % 64
    mask = torch.ones(B, N, dtype=torch.bool)
    mask[1, 3:] = False
    batch = types.SimpleNamespace(user_ids=torch.arange(B).cuda(), ids=item.cuda(), ids_fut=torch.tensor([[7], [8], [9], [10]]).cuda(),
                                  x=x[item].cuda(), x_fut=x[[7, 8, 9, 10]].cuda(), seq_mask=mask.cuda())
    out_live = tok(batch)  # no cache yet: encodes batch.x
    tok.precompute_corpus_ids(x)
    out_cached = tok(batch)
    exp = want[item.numpy()].reshape(B, N * 3)
    exp[~np.repeat(mask.numpy(), 3, axis=1)] = -1
    for out in (out_live, out_cached):
        assert np.array_equal(out.sem_ids.cpu().numpy(), exp)
        assert np.array_equal(out.sem_ids_fut.cpu().numpy(), want[[7, 8, 9, 10]])
        assert out.token_type_ids.shape == (B, N * 3) and out.token_type_ids_fut.shape == (B, 3)
        assert out.seq_mask.shape == (B, N * 3)


@pytest.mark.parametrize("mode", ["concat", "inter"])
def test_combined_id_modes(mode):
    tok, cfg, P = make_tok(mode)
    x, _, _ = O.formula_batch(cfg, 40, seed=33, tagged=False)
    ids = tok.precompute_corpus_ids(x).cpu().numpy()
    sem = oracle_ids(cfg, P, x)
    tok.hrq_vae.eval()  # (the tokenizer's eval_mode wrapper restores train(True) on every child, as the reference's does)
    tags = tok.hrq_vae.predict_tags(x.cuda())["predictions"].cpu().numpy()
    assert tok.sem_ids_dim == 6 and ids.shape == (40, 6)
    if mode == "concat":
        assert np.array_equal(ids, np.concatenate([sem, tags], 1))
    else:
        assert np.array_equal(ids[:, 0::2], sem) and np.array_equal(ids[:, 1::2], tags)
    # eval-mode tag predictions agree with the oracle's tag heads
    with torch.no_grad():
        z = O.mlp(x, O.enc_weights(P, cfg), True)
        res, embs = z, []
        for i in range(3):
            o, _, _, _ = O.quantize_level(res, O.effective_codebook(P, cfg, i), cfg.codebook_mode, 0.25, False, 0.2, None)
            embs.append(o)
            res = res - o
            Pe = dict(P)
            logits = O.tag_predictor(Pe, cfg, i, torch.cat(embs, -1), False, None)
            assert np.array_equal(tags[:, i], logits.argmax(-1).numpy())
