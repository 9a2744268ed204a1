"""GPU: the reference's small modules as stand-alone callables on the HIP path -- ReconstructionLoss / QuantizeLoss
(reference modules/loss.py:7-12, 36-44), SemanticIdUniquenessLoss called directly (h_rqvae.py:41-105), distributions/gumbel.py --
each against the torch expression the reference evaluates, values and gradients."""
import types

import numpy as np
import pytest
import torch

from oracle import fill
from tests import helpers as H

pytestmark = pytest.mark.gpu


def dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(grad)


@pytest.mark.parametrize("shape", [(64, 768), (5, 7, 32), (1, 3)])
def test_reconstruction_loss_module(shape):
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.loss import ReconstructionLoss
    a, b = fill.gauss(shape, 11), fill.gauss(shape, 12)
    xa, xb = dev(a, True), dev(b, True)
    out = ReconstructionLoss()(xa, xb)
    ra, rb = torch.from_numpy(a).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    ref = ((ra - rb) ** 2).sum(axis=-1)  # loss.py:12
    assert out.shape == ref.shape and H.rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-6
    w = torch.from_numpy(fill.gauss(tuple(ref.shape), 13))
    (out * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    assert H.close(xa.grad.cpu().numpy(), ra.grad.numpy(), 1e-6, 1e-9) and H.close(xb.grad.cpu().numpy(), rb.grad.numpy(), 1e-6, 1e-9)


@pytest.mark.parametrize("cw", [0.25, 0.4, 1.0])
def test_quantize_loss_module(cw):
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.loss import QuantizeLoss
    q, v = fill.gauss((96, 32), 21), fill.gauss((96, 32), 22)
    xq, xv = dev(q, True), dev(v, True)
    out = QuantizeLoss(cw)(xq, xv)
    rq, rv = torch.from_numpy(q).requires_grad_(True), torch.from_numpy(v).requires_grad_(True)
    ref = ((rq.detach() - rv) ** 2).sum(axis=[-1]) + cw * ((rq - rv.detach()) ** 2).sum(axis=[-1])  # loss.py:41-44
    assert H.rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-6
    out.sum().backward()
    ref.sum().backward()
    assert H.close(xq.grad.cpu().numpy(), rq.grad.numpy(), 1e-6, 1e-9) and H.close(xv.grad.cpu().numpy(), rv.grad.numpy(), 1e-6, 1e-9)


def test_uniqueness_loss_module_is_differentiable_like_the_references():
    """SemanticIdUniquenessLoss.forward(sem_ids [n,m], encoded_features) composed directly (as reference code does under
    install_dropin): rows of the id matrix that agree everywhere are pushed apart; value and d/d encoded_features vs the
    reference's expression (h_rqvae.py:52-105)."""
    import hidvae_amd  # noqa: F401
    from hidvae_amd.modules.h_rqvae import SemanticIdUniquenessLoss
    n, m = 6, 4
    ids = torch.tensor([[1, 2, 3, 4], [1, 2, 3, 4], [5, 5, 5, 5], [1, 2, 3, 4], [5, 5, 5, 5], [9, 9, 9, 9]])
    z = fill.gauss((n + 3, 32), 31)  # more feature rows than id rows: the loss touches rows [0, n) only
    margin, weight = 0.1, 1.5
    xz = dev(z, True)
    out = SemanticIdUniquenessLoss(margin=margin, weight=weight)(ids.cuda(), xz)
    rz = torch.from_numpy(z).requires_grad_(True)
    eq = (ids[:, None, :] == ids[None, :, :]).all(-1) & torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
    a, b = torch.nonzero(eq, as_tuple=True)
    cos = torch.nn.functional.cosine_similarity(rz[a], rz[b], dim=1)
    ref = weight * torch.relu(cos - margin).mean()
    assert abs(float(out.detach()) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert out.requires_grad
    out.backward()
    ref.backward()
    assert H.close(xz.grad.cpu().numpy(), rz.grad.numpy(), 1e-5, 1e-8)
    assert not xz.grad[n:].any()


def test_gumbel_module_functions():
    import hidvae_amd  # noqa: F401
    from hidvae_amd import _C
    from hidvae_amd.distributions.gumbel import TemperatureScheduler, gumbel_softmax_sample, sample_gumbel
    U = torch.from_numpy(fill.uniform((33, 100), 41, 0.0, 1.0))
    got = _C.gumbel_noise(U.cuda()).cpu()
    want = -torch.log(-torch.log(U + 1e-20) + 1e-20)  # distributions/gumbel.py:11
    assert H.close(got.numpy(), want.numpy(), 2e-6, 1e-6)
    logits = torch.from_numpy(fill.gauss((33, 100), 42))
    got = _C.gumbel_softmax_rows(logits.cuda(), U.cuda(), 0.7).cpu()
    want = torch.softmax((logits + want) / 0.7, dim=-1)  # :16-17
    assert H.close(got.numpy(), want.numpy(), 1e-5, 1e-8) and H.rel_err(got.sum(-1).numpy(), np.ones(33)) < 1e-6
    g = sample_gumbel((8, 16), torch.device("cuda"))
    s = gumbel_softmax_sample(torch.zeros(8, 16, device="cuda"), 0.5, torch.device("cuda"))
    assert g.shape == (8, 16) and torch.isfinite(g).all() and H.rel_err(s.sum(-1).cpu().numpy(), np.ones(8)) < 1e-6
    ts = TemperatureScheduler(1.0, 0.1, 1e-3, 10)
    assert ts.get_t(0) == 1.0 and ts.get_t(9) < 1.0


def test_reference_checkpoint_loads_through_load_pretrained(tmp_path):
    """a checkpoint in the reference's layout written by save_checkpoint (torch-format optimizer state, the enum pickled under the
    reference's module path) comes back through HRqVae.load_pretrained + HidvaeAdamW.load_state_dict on the device, and the resumed
    optimizer takes the same next step as the one that was saved"""
    from oracle import torch_oracle as O
    from hidvae_amd.checkpoint import load_checkpoint, save_checkpoint
    from hidvae_amd.optim import HidvaeAdamW
    from tests.test_model_gpu import build_model, make_batch
    cfg = O.Cfg(commitment_weight=0.4)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    x, _, _ = O.formula_batch(cfg, 64, seed=7, tagged=False)

    def core_groups(m):
        return [{"params": list(m.encoder.parameters()) + list(m.decoder.parameters()), "lr": 2.8e-4, "weight_decay": 0.015},
                {"params": [p for layer in m.layers for p in layer.parameters()], "lr": 2.8e-4, "weight_decay": 0.015}]

    def one_step(m, opt):
        opt.zero_grad()
        m(make_batch(x, None, None), gumbel_t=0.2).loss.backward()
        opt.step()

    m = build_model(cfg, P).train()
    opt = HidvaeAdamW(core_groups(m), cosine=(1000, 7e-8))
    for _ in range(3):
        one_step(m, opt)
    path = str(tmp_path / "ck.pt")
    save_checkpoint({"iter": 2, "model": m.state_dict(), "model_config": m.config, "optimizer": opt.state_dict(), "accuracy": 0.7,
                     "rqvae_loss": 0.3, "sem_id_repetition_rate": 0.01}, path)
    one_step(m, opt)  # the step the resumed run must reproduce
    m2 = build_model(cfg, O.formula_params(cfg, seed=5, with_tags=True)).train()
    m2.load_pretrained(path)
    opt2 = HidvaeAdamW(core_groups(m2), cosine=(1000, 7e-8), start_step=3)
    assert opt2.load_state_dict(load_checkpoint(path, map_location="cuda")["optimizer"]) is True
    one_step(m2, opt2)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("B,w", [(5000, 64), (4096, 96)])
def test_chunked_infonce_equals_the_materialised_form(B, w, monkeypatch):
    """From 4096 items on InfoNCE (reference loss.py:54-85) runs on column chunks of the similarity matrix with an online
    logsumexp and a recomputing backward; it must agree with the form that materialises softmax(S / tau): loss to 1e-6, gradients to
    1e-5 of their maximum (different summation order of the softmax denominators)."""
    import hidvae_amd  # noqa: F401
    from hidvae_amd import tagpath
    c0, t0 = fill.gauss((B, w), 61), fill.gauss((B, w), 62)

    def run(chunk_from):
        monkeypatch.setattr(tagpath, "INFONCE_CHUNK_FROM", chunk_from)
        c, t = dev(c0, True), dev(t0, True)
        loss = tagpath.InfoNCEFn.apply(c, t, 0.1, 0.15)
        (loss * 3.0).backward()
        return float(loss.detach()), c.grad.cpu().numpy(), t.grad.cpu().numpy()

    l1, gc1, gt1 = run(1)          # chunked
    l0, gc0, gt0 = run(10 ** 9)    # materialised
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    assert H.close(gc1, gc0, 1e-5, 1e-10) and H.close(gt1, gt0, 1e-5, 1e-10)
    # and against the reference's expression in float64
    cn = torch.nn.functional.normalize(torch.from_numpy(c0).double(), dim=-1)
    tn = torch.nn.functional.normalize(torch.from_numpy(t0).double(), dim=-1)
    ref = 0.15 * torch.nn.functional.cross_entropy(cn @ tn.T / 0.1, torch.arange(B))
    assert abs(l1 - float(ref)) <= 2e-6 * abs(float(ref))


def test_warm_up_forward_at_20000_tagged_items_stays_under_a_gigabyte():
    """train_hidvae.py:692-696: the k-means warm-up is one training-mode forward over min(20000, N) items.  With the tag heads that
    used to materialise three B x B softmax matrices (1.6 GB each); now the whole pass peaks below 1 GB beyond its inputs."""
    import types
    from oracle import torch_oracle as O
    from tests.test_model_gpu import build_model
    cfg = O.Cfg(commitment_weight=0.4, tag_class_counts=[38, 168, 348])
    m = build_model(cfg, O.formula_params(cfg, seed=100, with_tags=True)).train()
    B = 20000
    g = torch.Generator(device="cuda").manual_seed(0)
    batch = types.SimpleNamespace(x=torch.nn.functional.normalize(torch.randn(B, 768, device="cuda", generator=g), dim=-1),
                                  tags_emb=torch.randn(B, 3, 768, device="cuda", generator=g),
                                  tags_indices=torch.randint(0, 38, (B, 3), device="cuda", generator=g))
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        out = m(batch, gumbel_t=0.2)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert torch.isfinite(out.loss) and float(out.tag_align_loss) > 0
    assert peak < 1 << 30, f"warm-up forward peaked {peak / 2**20:.0f} MiB above its inputs"


@pytest.mark.parametrize("E,normalize,B", [(32, False, 37), (64, True, 128), (96, True, 1000), (128, True, 9)])
def test_gate_launch_matches_the_layer_by_layer_gate(E, normalize, B):
    """TagPredictor's attention gate as ONE row-local launch each way (hidvae_gate_fwd / hidvae_gate_bwd + one grouped launch for the
    weight gradients) against the composition it replaces (three Linear launches with ReLU / GELU / sigmoid epilogues, the product,
    F.normalize -- and autograd through them) and against float64 torch: reference h_rqvae.py:128-139, 196-206."""
    from hidvae_amd.ops import L2NormFn, LinearFn
    from hidvae_amd.tagpath import GateFn, MulFn
    from hidvae_amd import _C
    g = torch.Generator().manual_seed(E + B)
    mk = lambda *s_: (torch.randn(*s_, generator=g) * 0.3).cuda().requires_grad_()
    cat = torch.randn(B, 128, generator=g).cuda()  # x is a column-prefix VIEW of a wider buffer, as in the step
    W0, b0, W2, b2, W4, b4 = mk(E // 4, E), mk(E // 4), mk(E // 2, E // 4), mk(E // 2), mk(E, E // 2), mk(E)
    gout = torch.randn(B, E, generator=g).cuda()

    def run(fused):
        for t in (W0, b0, W2, b2, W4, b4):
            t.grad = None
        x = cat[:, :E].detach().requires_grad_()  # (a view with row stride 128)
        xv = x
        if fused:
            h = GateFn.apply(xv, W0, b0, W2, b2, W4, b4, normalize)
        else:
            a = LinearFn.apply(LinearFn.apply(LinearFn.apply(xv, W0, b0, _C.EPI_RELU), W2, b2, _C.EPI_GELU), W4, b4, _C.EPI_SIGMOID)
            h = MulFn.apply(xv, a)
            if normalize:
                h = L2NormFn.apply(h, 1e-12)
        h.backward(gout)
        return h.detach(), x.grad.clone(), [t.grad.clone() for t in (W0, b0, W2, b2, W4, b4)]

    hf, gxf, gpf = run(True)
    hu, gxu, gpu = run(False)
    # float64 reference
    x64 = cat[:, :E].double().cpu().requires_grad_()
    P64 = [t.detach().double().cpu().requires_grad_() for t in (W0, b0, W2, b2, W4, b4)]
    a1 = torch.relu(x64 @ P64[0].t() + P64[1])
    a2 = torch.nn.functional.gelu(a1 @ P64[2].t() + P64[3])
    a3 = torch.sigmoid(a2 @ P64[4].t() + P64[5])
    h64 = x64 * a3
    if normalize:
        h64 = torch.nn.functional.normalize(h64, dim=-1, eps=1e-12)
    h64.backward(gout.double().cpu())
    close = lambda a, b, tol: float((a.double().cpu() - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(hf, h64.detach(), 2e-6) and close(hu, h64.detach(), 2e-6)
    assert close(gxf, x64.grad, 5e-6), float((gxf.double().cpu() - x64.grad).abs().max())
    for got, want in zip(gpf, P64):
        assert close(got, want.grad, 2e-5 * max(1.0, (B / 256) ** 0.5)), float((got.double().cpu() - want.grad).abs().max())
    assert close(gxf, gxu.double().cpu(), 5e-6)


@pytest.mark.parametrize("M,N,relu,in_gate", [(1000, 691, True, 0.0), (128, 768, False, 1.0 / 0.6), (37, 64, True, 2.0), (4096, 230, False, 0.0)])
def test_layernorm_backward_split_at_its_seam(M, N, relu, in_gate):
    """hidvae_layernorm_bwd_partial + hidvae_layernorm_param_final_many (one finishing launch for MANY LayerNorms) against the
    one-LayerNorm-at-a-time hidvae_layernorm_bwd_all: the ReLU -> Dropout gate read off the forward OUTPUT instead of the keep-mask
    gives the same bits; the input-side gate (in_relu_scale) equals hidvae_act_bwd applied to gx; the affine gradients are
    bit-identical (same partials, same summation order), also accumulated into an existing gradient."""
    from hidvae_amd import _C
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x = torch.randn(M, N, device="cuda", generator=g)
    if in_gate:
        x = torch.relu(x) * (torch.rand(M, N, device="cuda", generator=g) < 0.6) * in_gate  # a ReLU -> Dropout output
    gamma, beta = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    mask = (torch.rand(M, N, device="cuda", generator=g) < 0.6).float() if relu else None
    scale = 1.0 / 0.6 if relu else 1.0
    y, mean, rstd = _C.layernorm_fwd(x, gamma, beta, 1e-5, relu, mask, scale, None)
    gy = torch.randn(M, N, device="cuda", generator=g)
    gx0, gg0, gb0 = _C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, scale)
    if in_gate:
        gx0 = _C.act_bwd(gx0, x, _C.EPI_RELU, x, in_gate)  # (mask pointer only switches the scale on: x > 0 is the gate)
    gx1, part = _C.layernorm_bwd_partial(gy, x, gamma, beta, mean, rstd, relu, y if relu else None, scale, in_gate)
    gg1, gb1 = torch.full((N,), 3.0, device="cuda"), torch.full((N,), -2.0, device="cuda")
    other = torch.randn(8, 40, device="cuda", generator=g)  # a second, unrelated LayerNorm finished by the same launch
    oy, om, orr = _C.layernorm_fwd(other, torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"), 1e-5, False, None, 1.0, None)
    ogy = torch.randn(8, 40, device="cuda", generator=g)
    _, opart = _C.layernorm_bwd_partial(ogy, other, torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"), om, orr, False, None, 1.0, 0.0)
    ogg, ogb = torch.empty(40, device="cuda"), torch.empty(40, device="cuda")
    _C.layernorm_param_final_many([(part, M, N, gg1, gb1, True), (opart, 8, 40, ogg, ogb, False)])
    assert torch.equal(gx1, gx0)
    assert torch.equal(gg1, gg0 + 3.0) and torch.equal(gb1, gb0 - 2.0)
    _, wg, wb = _C.layernorm_bwd_all(ogy, other, torch.ones(40, device="cuda"), torch.zeros(40, device="cuda"), om, orr, False, None, 1.0)
    assert torch.equal(ogg, wg) and torch.equal(ogb, wb)


@pytest.mark.parametrize("name", H.case_names("quantize"))
def test_quantize_level_with_the_cosine_ranking_matches_the_reference(name):
    """Quantize(distance_mode=COSINE) on its own (reference modules/quantize.py:115-119): ids bit-exact, embeddings / loss / both
    gradients against the reference's own run; the launch itself bit for bit against ORDER-GEN of oracle/exact.c."""
    from hidvae_amd import _C
    from hidvae_amd.modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode
    from oracle import exact
    fx, d = H.load(name)
    x, E, g_out, g_loss = H.quantize_inputs(d)
    q = Quantize(embed_dim=d["D"], n_embed=d["K"], do_kmeans_init=False, codebook_normalize=d["normalize"], commitment_weight=d["beta"],
                 forward_mode=QuantizeForwardMode(d["mode"]), distance_mode=QuantizeDistance.COSINE).cuda()
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(E))
    q.train(d["training"])
    xt = torch.from_numpy(x).cuda().requires_grad_(d["training"])
    out = q(xt, temperature=0.2)
    assert np.array_equal(out.ids.cpu().numpy(), fx["ids"].astype(np.int64))
    assert H.rel_err(out.embeddings.detach().cpu().numpy(), fx["embeddings"]) <= 1e-5
    assert H.rel_err(out.loss.detach().cpu().numpy(), fx["loss"]) <= 1e-5
    if d["training"]:
        ((out.embeddings * torch.from_numpy(g_out).cuda()).sum() + (out.loss * torch.from_numpy(g_loss).cuda()).sum()).backward()
        assert H.close(xt.grad.cpu().numpy(), fx["grad_x"], 2e-5, 1e-7)
        assert H.close(q.embedding.weight.grad.cpu().numpy(), fx["grad_E"], 2e-5, 1e-7)
    want = exact.rq_forward(x, [E], False, d["normalize"], d["mode"], d["training"], d["beta"], cosine=True)
    cb, cc = _C.codebook_prepare([torch.from_numpy(E).cuda()], [d["normalize"]])
    _, ids, emb_cat, _, _, qloss = _C.rq_forward(torch.from_numpy(x).cuda(), cb, cc, False, d["mode"], d["training"], d["beta"],
                                                 distance=_C.DIST_COSINE)
    assert np.array_equal(ids.cpu().numpy(), want["ids"])
    assert np.array_equal(emb_cat.cpu().numpy(), want["emb_cat"])
    assert np.array_equal(qloss.cpu().numpy(), want["loss"])


@pytest.mark.parametrize("name", H.case_names("qgumbel"))
def test_quantize_level_gumbel_softmax_with_the_cosine_ranking_matches_the_reference(name):
    """Quantize(GUMBEL_SOFTMAX -- the ctor default --, distance_mode=COSINE) in training (reference modules/quantize.py:115-119,125-130),
    D = 32 and 64, with and without codebook normalisation, the Gumbel draws replayed from the fixture's formula: ids on the rows whose
    top-2 gap is not a near-tie, embeddings / loss / both gradients against the reference's own run"""
    from hidvae_amd.modules.quantize import Quantize, QuantizeDistance, QuantizeForwardMode
    from oracle import fill
    fx, d = H.load(name)
    x, E, g_out, g_loss = H.quantize_inputs(d)
    U = fill.uniform((d["B"], d["K"]), d["seed"] + 5, 0.0, 1.0)
    q = Quantize(embed_dim=d["D"], n_embed=d["K"], do_kmeans_init=False, codebook_normalize=d["normalize"], commitment_weight=d["beta"],
                 forward_mode=QuantizeForwardMode.GUMBEL_SOFTMAX, distance_mode=QuantizeDistance.COSINE).cuda()
    q.rand = types.SimpleNamespace(gumbel_u=lambda shape, device: torch.from_numpy(U).to(device).reshape(tuple(shape)))
    with torch.no_grad():
        q.embedding.weight.copy_(torch.from_numpy(E))
    q.train(True)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    out = q(xt, temperature=d["temperature"])
    safe = fx["margins"] > 1e-6
    assert np.array_equal(out.ids.cpu().numpy()[safe], fx["ids"].astype(np.int64)[safe])
    assert H.rel_err(out.embeddings.detach().cpu().numpy(), fx["embeddings"]) <= 2e-5
    assert H.rel_err(out.loss.detach().cpu().numpy(), fx["loss"]) <= 2e-5
    ((out.embeddings * torch.from_numpy(g_out).cuda()).sum() + (out.loss * torch.from_numpy(g_loss).cuda()).sum()).backward()
    assert H.close(xt.grad.cpu().numpy(), fx["grad_x"], 5e-5, 1e-6)
    assert H.close(q.embedding.weight.grad.cpu().numpy(), fx["grad_E"], 5e-5, 1e-6)


def test_softmax_argmax_rows_is_torchs_softmax_max():
    """HRqVae.predict_tags' launch (reference h_rqvae.py:716-722: torch.softmax(logits, -1).max(-1)): first arg max per row and its
    softmax probability, written into one column of the stacked [B, L] outputs; ragged widths, ties, one-column logits"""
    from hidvae_amd import _C
    g = torch.Generator(device="cuda").manual_seed(5)
    for B, C in ((1, 1), (5, 38), (333, 168), (1024, 348), (64, 1000)):
        logits = torch.randn(B, C, device="cuda", generator=g) * 3
        if C > 4:
            logits[0, 3] = logits[0, 1] = logits[0].max() + 1.0  # a tie: the first maximal column wins
        pred = torch.full((B, 3), -7, dtype=torch.int64, device="cuda")
        conf = torch.full((B, 3), -7.0, device="cuda")
        _C.softmax_argmax_rows(logits, pred, conf, 1)
        want_c, _ = torch.softmax(logits.double(), dim=-1).max(dim=-1)
        want_p = torch.from_numpy(np.argmax(logits.cpu().numpy(), axis=1))  # (numpy: first maximum)
        assert torch.equal(pred[:, 1].cpu(), want_p)
        assert torch.allclose(conf[:, 1].double(), want_c, rtol=2e-6, atol=1e-7)
        assert (pred[:, 0] == -7).all() and (pred[:, 2] == -7).all() and (conf[:, 0] == -7).all() and (conf[:, 2] == -7).all()


@pytest.mark.parametrize("E,H,C,layer_idx,B,p", [(32, 256, 38, 0, 1024, 0.4), (64, 128, 17, 1, 100, 0.25), (96, 250, 348, 2, 37, 0.0)])
def test_one_launch_predictor_forward_matches_the_separate_launches(E, H, C, layer_idx, B, p):
    """hidvae_predictor_fwd (TagPredictor behind its gate as ONE row-local launch, reference h_rqvae.py:132-188) against the launch
    sequence it replaces, under the SAME in-kernel dropout decisions: the logits, the input gradient and every parameter gradient
    (the backward is the same Functions either way; only what they were handed differs in summation order of the LayerNorms)."""
    import os
    from hidvae_amd.modules.h_rqvae import TagPredictor
    from hidvae_amd.rand import DeviceRand
    from hidvae_amd.tagpath import tag_predictor_forward, flush_layernorm_finals
    torch.manual_seed(E * 1000 + H)
    pred = TagPredictor(E, C, hidden_dim=H, dropout_rate=p, use_batch_norm=True, layer_idx=layer_idx).cuda().train()
    with torch.no_grad():
        for q in pred.parameters():
            if q.dim() == 1:
                q.add_(torch.randn_like(q) * 0.1)  # (biases / affine parameters off their initial 0 / 1)
    g = torch.Generator().manual_seed(B + C)
    cat = torch.randn(B, 128, generator=g).cuda()
    gout = torch.randn(B, C, generator=g).cuda()

    def run(fused, bwd=True):
        os.environ["HIDVAE_FUSED_PREDICTOR"] = "1" if fused else "0"
        os.environ["HIDVAE_FUSED_PREDICTOR_BWD"] = "1" if bwd else "0"
        try:
            for q in pred.parameters():
                q.grad = None
            rand = DeviceRand(seed=77)
            rand.begin_step(cat.device)
            x = cat[:, :E].detach().requires_grad_()
            logits = tag_predictor_forward(pred, x, None, rand)
            logits.backward(gout)
            flush_layernorm_finals()
            torch.cuda.synchronize()
            return logits.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in pred.named_parameters()}
        finally:
            os.environ.pop("HIDVAE_FUSED_PREDICTOR", None)
            os.environ.pop("HIDVAE_FUSED_PREDICTOR_BWD", None)

    lu, gxu, gpu = run(False)
    rel = lambda a, b: float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))
    for bwd in (False, True):  # the one-launch forward under the separate backward launches, then with the one-launch backward too
        lf, gxf, gpf = run(True, bwd)
        assert rel(lf, lu) <= 1e-5, (bwd, rel(lf, lu))
        assert rel(gxf, gxu) <= 2e-5, (bwd, rel(gxf, gxu))
        for n in gpu:
            assert rel(gpf[n], gpu[n]) <= 3e-5 * max(1.0, (B / 256) ** 0.5), (bwd, n, rel(gpf[n], gpu[n]))
    if p > 0:  # the two runs really dropped the same units: an activation that is exactly zero in one is zero in the other
        assert float((lf - lu).abs().max()) < 1e-3
