"""GPU: every C-ABI kernel of the core path against the oracle (bit-exact where the order is specified)."""
import numpy as np
import pytest
import torch

from oracle import exact, fill, torch_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    import hidvae_amd
    from hidvae_amd import _C
    _C.lib()
    return _C


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (1024, 512, 768), (128, 32, 128), (100, 38, 115), (33, 230, 691),
                                   (7, 5, 3), (256, 768, 512), (4096, 1024, 96), (8192, 512, 40),
                                   (1000, 500, 333), (2048, 512, 777)])  # (the last two: gemm_tile16_kernel, ragged K, both tile shapes)
def test_gemm_nt_is_the_sequential_fmaf_chain(C, M, N, K):
    x = fill.uniform((M, K), 1, -1, 1)
    w = fill.uniform((N, K), 2, -1, 1)
    want = exact.linear(x, w)
    got = C.gemm(C.GEMM_NT, dev(x), dev(w)).cpu().numpy()
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_gemm_silu_epilogue_bit_exact(C):
    x = fill.uniform((200, 96), 3, -2, 2)
    w = fill.uniform((72, 96), 4, -1, 1)
    want = exact.linear(x, w, silu=True)
    aux = torch.empty((200, 72), device="cuda")
    got = C.gemm(C.GEMM_NT, dev(x), dev(w), epilogue=C.EPI_SILU, aux=aux).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(aux.cpu().numpy(), exact.linear(x, w))


@pytest.mark.parametrize("layout", ["NN", "TN"])
@pytest.mark.parametrize("M,N,K,split", [(96, 80, 72, 1), (130, 67, 45, 1), (512, 768, 1024, 0), (32, 128, 1000, 0),
                                         (1024, 512, 768, 0), (4096, 1024, 200, 4), (8, 8, 5000, 0)])
def test_gemm_other_layouts(C, layout, M, N, K, split):
    a = fill.uniform((M, K), 5, -1, 1)
    b = fill.uniform((K, N), 6, -1, 1)
    want = a.astype(np.float64) @ b.astype(np.float64)
    if layout == "NN":
        got = C.gemm(C.GEMM_NN, dev(a), dev(b), split_k=split)
    else:
        got = C.gemm(C.GEMM_TN, dev(np.ascontiguousarray(a.T)), dev(b), split_k=split)
    assert H.rel_err(got.cpu().numpy(), want) < 2e-6


def test_gemm_views_bias_accumulate_and_backward_epilogue(C):
    big = dev(fill.uniform((50, 96), 7, -1, 1))
    a = big[:, :64]  # column-slice view, lda = 96
    w = dev(fill.uniform((40, 64), 8, -1, 1))
    bias = dev(fill.uniform((40,), 9, -1, 1))
    out = torch.ones((50, 40), device="cuda")
    C.gemm(C.GEMM_NT, a, w, out=out, bias=bias, accumulate=True)
    want = 1.0 + a.double() @ w.double().T + bias.double()
    assert H.rel_err(out.cpu().numpy(), want.cpu().numpy()) < 2e-6
    pre = dev(fill.uniform((50, 40), 10, -3, 3))
    g = C.gemm(C.GEMM_NT, a, w, epilogue=C.EPI_DSILU, aux=pre)
    s = torch.sigmoid(pre.double())
    want = (a.double() @ w.double().T) * (s * (1 + pre.double() * (1 - s)))
    assert H.rel_err(g.cpu().numpy(), want.cpu().numpy()) < 2e-6


def test_colsum(C):
    x = dev(fill.uniform((1000, 230), 11, -1, 1))
    assert H.rel_err(C.colsum(x).cpu().numpy(), x.double().sum(0).cpu().numpy()) < 2e-6


def _rq_inputs(B, L, K, seed, scale=1.0, D=32):
    y = fill.gauss((B, D), seed) * np.float32(scale)
    tables = [fill.uniform((K, D), seed + 1 + i, -1, 1) * np.float32(1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
    return y, tables


@pytest.mark.parametrize("B,L,K", [(1, 1, 1), (16, 3, 256), (1000, 3, 256), (77, 4, 1024), (130, 2, 40), (64, 8, 16),
                                   (300, 1, 2048), (4096, 3, 256)])
@pytest.mark.parametrize("mode,training", [(3, True), (2, True), (3, False)])
def test_rq_forward_bit_exact_vs_c_oracle(C, B, L, K, mode, training):
    y, tables = _rq_inputs(B, L, K, 20 + B % 7)
    want = exact.rq_forward(y, tables, True, True, mode, training, 0.4)
    cb, cc = C.codebook_prepare([dev(t) for t in tables], [i == 0 for i in range(L)])
    assert np.array_equal(cb.cpu().numpy(), np.stack(want["cbs"]))
    assert np.array_equal(cc.cpu().numpy(), np.stack(want["ccs"]))
    z, ids, emb_cat, emb_sum, res, qloss = C.rq_forward(dev(y), cb, cc, True, mode, training, 0.4, want_res=True)
    assert np.array_equal(ids.cpu().numpy(), want["ids"]), "semantic ids differ"
    for got, key in ((z, "z"), (emb_cat, "emb_cat"), (emb_sum, "emb_sum"), (res, "res_cat"), (qloss, "loss")):
        assert np.array_equal(got.cpu().numpy(), want[key]), key


@pytest.mark.parametrize("D", [64, 16, 48, 4])
@pytest.mark.parametrize("B,L,K", [(1, 1, 1), (130, 3, 256), (77, 4, 1000), (1027, 2, 40)])
@pytest.mark.parametrize("mode,training", [(3, True), (2, True), (3, False)])
def test_rq_forward_other_widths_bit_exact_vs_c_oracle(C, D, B, L, K, mode, training):
    """embed_dim other than 32 (configs/rqvae_ml32m.gin:11 uses 64): csrc/rq_generic.hip against ORDER-GEN of oracle/exact.c."""
    y, tables = _rq_inputs(B, L, K, 60 + D, D=D)
    want = exact.rq_forward(y, tables, True, True, mode, training, 0.4)
    cb, cc = C.codebook_prepare([dev(t) for t in tables], [i == 0 for i in range(L)])
    assert np.array_equal(cb.cpu().numpy(), np.stack(want["cbs"]))
    assert np.array_equal(cc.cpu().numpy(), np.stack(want["ccs"]))
    z, ids, emb_cat, emb_sum, res, qloss = C.rq_forward(dev(y), cb, cc, True, mode, training, 0.4, want_res=True)
    assert np.array_equal(ids.cpu().numpy(), want["ids"]), "semantic ids differ"
    for got, key in ((z, "z"), (emb_cat, "emb_cat"), (emb_sum, "emb_sum"), (res, "res_cat"), (qloss, "loss")):
        assert np.array_equal(got.cpu().numpy(), want[key]), key
    if not training:  # the ids-only launch is the eval-mode search
        assert np.array_equal(C.rq_ids(dev(y), cb, cc, True).cpu().numpy(), want["ids"])


def test_embed_dim_outside_the_supported_set_is_refused(C):
    for D in (0, 6, 68, 128):
        with pytest.raises((RuntimeError, ValueError)):
            C.check_embed_dim(D)
    for D in (4, 16, 32, 64):
        C.check_embed_dim(D)


def test_rq_exact_ties_pick_lowest_index(C):
    y = np.zeros((70, 32), np.float32)
    y[:, 0] = 1.0
    E = np.zeros((48, 32), np.float32)
    E[:, 1] = 1.0
    cb, cc = C.codebook_prepare([dev(E), dev(E.copy())], [False, False])
    _, ids, *_ = C.rq_forward(dev(y), cb, cc, False, 2, True, 0.25)
    assert int(ids.abs().sum()) == 0
    E2 = E.copy()
    E2[37] = y[0]  # a single exact match must win wherever it sits
    cb, cc = C.codebook_prepare([dev(E2)], [False])
    _, ids, *_ = C.rq_forward(dev(y), cb, cc, False, 2, True, 0.25)
    assert (ids.cpu().numpy() == 37).all()


@pytest.mark.parametrize("name", [n for n in H.case_names("case") if "gumbel" not in n and "simvq" not in n])
def test_rq_forward_on_reference_goldens(C, name):
    """Reference z (from the fixture) -> HIP RQ -> the reference's own ids / embeddings / losses."""
    fx, desc = H.load(name)
    if "z" not in fx:
        pytest.skip("fixture stores no z")
    cfg, P, *_ = H.inputs_of(desc)
    tables = [P[f"layers.{i}.embedding.weight"].cuda() for i in range(cfg.n_layers)]
    cb, cc = C.codebook_prepare(tables, [cfg.codebook_normalize and i == 0 for i in range(cfg.n_layers)])
    z, ids, emb_cat, emb_sum, res, qloss = C.rq_forward(dev(fx["z"]), cb, cc, False, cfg.codebook_mode, desc["training"],
                                                        cfg.commitment_weight, want_res=True)
    assert (fx["margins"] > 1e-6).all()
    assert np.array_equal(ids.cpu().numpy(), fx["sem_ids"].astype(np.int64))
    L = cfg.n_layers
    assert H.rel_err(emb_cat.cpu().numpy().reshape(-1, L, cfg.embed_dim).transpose(0, 2, 1), fx["embeddings"]) <= 1e-5
    assert H.rel_err(res.cpu().numpy().reshape(-1, L, cfg.embed_dim).transpose(0, 2, 1), fx["residuals"]) <= 1e-5
    assert H.rel_err(qloss.cpu().numpy(), fx["rqvae_loss"]) <= 1e-5


@pytest.mark.parametrize("mode", [3, 2])
@pytest.mark.parametrize("B,L,K,norm,D", [(50, 3, 256, True, 32), (200, 4, 64, False, 32), (17, 1, 16, True, 32), (5000, 3, 256, True, 32),
                                          (4100, 2, 6, False, 32), (300, 3, 256, True, 64), (129, 2, 40, False, 64), (70, 3, 33, True, 16)])
def test_rq_backward_vs_autograd(C, mode, B, L, K, norm, D):
    y, tables = _rq_inputs(B, L, K, 31, D=D)
    g_cat = fill.uniform((B, L * D), 32, -1, 1)
    g_sum = fill.uniform((B, D), 33, -1, 1)
    g_z = fill.uniform((B, D), 34, -1, 1)
    gq = fill.uniform((B,), 35, 0.1, 1)
    # autograd reference (oracle) for sum(g_cat*emb) + sum(g_sum*emb.sum) + sum(g_z*z) + sum(gq*loss)
    yt = torch.from_numpy(y).requires_grad_(True)
    Et = [torch.from_numpy(t).requires_grad_(True) for t in tables]
    zt = torch.nn.functional.normalize(yt, dim=-1, eps=1e-12) if norm else yt
    res, embs, loss = zt, [], 0
    for i in range(L):
        cbt = torch.nn.functional.normalize(Et[i], dim=-1, eps=1e-12) if (norm and i == 0) else Et[i]
        o, ids_i, l, _ = O.quantize_level(res, cbt, mode, 0.4, True, 0.2, None)
        embs.append(o)
        loss = loss + l
        res = res - o
    emb_cat = torch.cat(embs, -1)
    obj = (emb_cat * torch.from_numpy(g_cat)).sum() + (sum(embs) * torch.from_numpy(g_sum)).sum() \
        + (zt * torch.from_numpy(g_z)).sum() + (loss * torch.from_numpy(gq)).sum()
    obj.backward()
    cb, cc = C.codebook_prepare([dev(t) for t in tables], [norm and i == 0 for i in range(L)])
    z, ids, *_ = C.rq_forward(dev(y), cb, cc, norm, mode, True, 0.4)
    g_y, dE = C.rq_backward(dev(y), z, cb, cc, norm, mode, 0.4, ids, dev(g_cat), dev(g_sum), dev(g_z), 1.0, dev(gq))
    assert H.close(g_y.cpu().numpy(), yt.grad.numpy(), 2e-5, 1e-7)
    gE = C.codebook_grad(ids, dE, [dev(t) for t in tables], cb, [norm and i == 0 for i in range(L)])
    for i in range(L):
        assert H.close(gE[i].cpu().numpy(), Et[i].grad.numpy(), 2e-5, 1e-7), i


def test_recon_and_l2norm(C):
    y = dev(fill.gauss((300, 768), 40))
    x = dev(fill.unit_rows((300, 768), 41))
    gs = dev(fill.uniform((300,), 42, 0.1, 1.0))
    yt = y.cpu().clone().requires_grad_(True)
    xh = torch.nn.functional.normalize(yt, dim=-1, eps=1e-12)
    rec = ((xh - x.cpu()) ** 2).sum(-1)
    (rec * gs.cpu()).sum().backward()
    recon, x_hat, g_y = C.recon_fwd_bwd(y, x, gscale_items=gs, want_xhat=True, want_grad=True)
    assert H.rel_err(recon.cpu().numpy(), rec.detach().numpy()) < 1e-5
    assert H.rel_err(x_hat.cpu().numpy(), xh.detach().numpy()) < 1e-5
    assert H.close(g_y.cpu().numpy(), yt.grad.numpy(), 2e-5, 1e-8)
    for n in (32, 96, 230):
        v = dev(fill.gauss((70, n), 43))
        out, norms = C.l2norm_fwd(v)
        vt = v.cpu().clone().requires_grad_(True)
        ot = torch.nn.functional.normalize(vt, dim=-1)
        g = dev(fill.gauss((70, n), 44))
        (ot * g.cpu()).sum().backward()
        assert H.rel_err(out.cpu().numpy(), ot.detach().numpy()) < 1e-6
        assert H.close(C.l2norm_bwd(g, out, norms).cpu().numpy(), vt.grad.numpy(), 1e-5, 1e-8)
    z = dev(fill.gauss((40, 32), 45))
    assert np.array_equal(C.l2norm_fwd(z)[0].cpu().numpy(),
                          exact.rq_forward(z.cpu().numpy(), [np.ones((1, 32), np.float32)], True, False, 2, False, 0.0)["z"])


@pytest.mark.parametrize("B,N,c", [(70, 768, 18), (33, 100, 1), (5, 40, 39), (300, 1030, 64)])
def test_categorical_decoder_tail_against_autograd(C, B, N, c):
    """n_cat_features > 0 (reference h_rqvae.py:610-613, loss.py:15-33): the row l2norm, the head's second l2norm, squared error on
    the head and BCE-with-logits on the last c columns, forward and gradient, against torch autograd on the CPU."""
    F = torch.nn.functional
    y = fill.gauss((B, N), 46) * np.float32(3.0)
    x = fill.unit_rows((B, N), 47)
    x[:, -c:] = (x[:, -c:] > 0).astype(np.float32)
    gs = fill.uniform((B,), 48, 0.1, 1.0)
    yt = torch.from_numpy(y).requires_grad_(True)
    u = F.normalize(yt, dim=-1, eps=1e-12)
    xh = torch.cat([F.normalize(u[:, :-c], dim=-1, eps=1e-12), u[:, -c:]], -1)
    xt = torch.from_numpy(x)
    rec = ((xh[:, :-c] - xt[:, :-c]) ** 2).sum(-1) + F.binary_cross_entropy_with_logits(xh[:, -c:], xt[:, -c:], reduction="none").sum(-1)
    (rec * torch.from_numpy(gs)).sum().backward()
    recon, x_hat, g_y = C.recon_fwd_bwd(dev(y), dev(x), gscale_items=dev(gs), want_xhat=True, want_grad=True, n_cat=c)
    assert H.rel_err(recon.cpu().numpy(), rec.detach().numpy()) < 1e-5
    assert H.rel_err(x_hat.cpu().numpy(), xh.detach().numpy()) < 1e-5
    assert H.close(g_y.cpu().numpy(), yt.grad.numpy(), 3e-5, 1e-8)
    # the stand-alone module on a given x_hat (CategoricalReconstructionLoss.forward)
    from hidvae_amd.modules.loss import CategoricalReconstructionLoss
    ht = xh.detach().clone().requires_grad_(True)
    ref = ((ht[:, :-c] - xt[:, :-c]) ** 2).sum(-1) + F.binary_cross_entropy_with_logits(ht[:, -c:], xt[:, -c:], reduction="none").sum(-1)
    (ref * torch.from_numpy(gs)).sum().backward()
    hd = xh.detach().cuda().requires_grad_(True)
    out = CategoricalReconstructionLoss(c)(hd, dev(x))
    (out * dev(gs)).sum().backward()
    assert H.rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    assert H.close(hd.grad.cpu().numpy(), ht.grad.numpy(), 2e-5, 1e-8)
    with pytest.raises(RuntimeError):
        C.recon_fwd_bwd(dev(y), dev(x), n_cat=N)


def test_l2norm_pair_launches_equal_the_single_launches(C):
    """two row normalisations (and their backward) in one launch each -- what TagAlignmentLoss issues -- against the one-problem entry
    points, bit for bit, incl. a strided first operand (a column prefix of emb_cat)"""
    base = dev(fill.gauss((300, 96), 91))
    c = base[:, :64]                      # row stride 96
    t = dev(fill.gauss((300, 40), 92))
    oc, nc, ot, nt = C.l2norm_fwd_pair(c, t)
    rc, rnc = C.l2norm_fwd(c.contiguous())
    rt, rnt = C.l2norm_fwd(t)
    assert torch.equal(oc, rc) and torch.equal(nc, rnc) and torch.equal(ot, rt) and torch.equal(nt, rnt)
    gc, gt = dev(fill.gauss((300, 64), 93)), dev(fill.gauss((300, 40), 94))
    xc, xt = C.l2norm_bwd_pair(gc, oc, nc, gt, ot, nt)
    assert torch.equal(xc, C.l2norm_bwd(gc, rc, rnc)) and torch.equal(xt, C.l2norm_bwd(gt, rt, rnt))
    with pytest.raises(RuntimeError):
        C.l2norm_fwd_pair(c, t[:10])


def test_id_stats(C):
    ids = torch.from_numpy(fill.ints((5000, 3), 50, 12)).cuda()
    emb = dev(fill.gauss((5000, 96), 51))
    norms, pu = C.id_stats(emb, ids)
    assert abs(float(pu) - float(O.p_unique_fast(ids.cpu()))) < 1e-7
    assert abs(float(O.p_unique(ids.cpu()[:600])) - float(C.id_stats(emb[:600], ids[:600])[1])) < 1e-7
    assert H.rel_err(norms.cpu().numpy(), emb.cpu().reshape(5000, 3, 32).norm(dim=-1).numpy()) < 1e-6
    # ONE caller-owned census table serves consecutive calls of one batch size (generation-tagged, never cleared)
    table = C.census_scratch(5000, "cuda")
    for seed, hi in ((52, 4), (53, 200), (54, 2), (55, 12), (56, 1)):
        ids = torch.from_numpy(fill.ints((5000, 3), seed, hi)).cuda()
        assert abs(float(C.id_stats(emb, ids, scratch=table)[1]) - float(O.p_unique_fast(ids.cpu()))) < 1e-7, (seed, hi)
    with pytest.raises(RuntimeError):
        C.id_stats(emb[:600], ids[:600], scratch=table)  # a table of another batch size is refused


@pytest.mark.parametrize("name", H.case_names("kmeans"))
def test_kmeans_matches_reference_golden(C, name):
    """Lloyd to convergence from the same seeding the reference was given (init/kmeans.py:34-77)."""
    from hidvae_amd.init.kmeans import Kmeans
    fx, desc = H.load(name)
    x = torch.from_numpy(fill.gauss((desc["N"], desc["D"]), desc["seed"])).cuda()
    init = fill.perm(desc["N"], desc["seed"] + 1)[: desc["K"]]
    out = Kmeans(k=desc["K"], init_indices=init).run(x)
    assert H.rel_err(out.centroids.cpu().numpy(), fx["centroids"]) <= 1e-5
    agree = (out.assignment.cpu().numpy() == fx["assignment"]).mean()
    assert agree == 1.0, f"assignment agreement {agree}"
    # a fixed point: one more iteration changes nothing
    again = Kmeans(k=desc["K"], init_indices=init, max_iters=1)
    again.init_indices = None
    c = out.centroids.clone()
    a2 = torch.empty(desc["N"], dtype=torch.int32, device="cuda")
    nxt, sc, sh = torch.empty_like(c), torch.empty(desc["K"], device="cuda"), torch.empty((), device="cuda")
    C.kmeans_iter(x, c, a2, torch.zeros(desc["K"], dtype=torch.int64, device="cuda"), nxt, sc, sh)
    assert float(sh) == 0.0


@pytest.mark.parametrize("B,n_out,n_in", [(1024, 256, 128), (1024, 32, 128), (1024, 128, 32), (96, 40, 24), (1000, 512, 256),
                                          (1024, 768, 512), (17, 5, 3)])
def test_linear_bwd_pair_is_bit_identical_to_two_gemms(C, B, n_out, n_in):
    """(the mid-size layers -- 1024 x 768 x 512 here -- take the balanced LDS-shared kernel, whose summation order is its own:
    tight tolerance against the two GEMMs instead of bit identity; tests/test_kernels_gpu.py::test_linear_bwd_balanced_* cover it)"""
    g = dev(fill.gauss((B, n_out), 70))
    x = dev(fill.gauss((B, n_in), 71))
    w = dev(fill.gauss((n_out, n_in), 72))
    pre = dev(fill.gauss((B, n_in), 73))
    balanced = C.workspace_bytes(C.WS_LINEAR_BWD_ZEROED, B, n_out, n_in, 0) > 0
    same = (lambda a, b: H.rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-6) if balanced else torch.equal
    for epi, aux in ((C.EPI_NONE, None), (C.EPI_DSILU, pre)):
        dW, dX = C.linear_bwd(g, x, w, True, epi, aux)
        assert same(dW, C.gemm(C.GEMM_TN, g, x, split_k=0))
        assert same(dX, C.gemm(C.GEMM_NN, g, w, epilogue=epi, aux=aux, split_k=0))
    dW, dX = C.linear_bwd(g, x, w, False)
    assert dX is None and same(dW, C.gemm(C.GEMM_TN, g, x, split_k=0))
    dW2, dX2, db = C.linear_bwd(g, x, w, True, bias=True)  # bias gradient from the same launch
    assert same(dW2, dW) and same(dX2, C.gemm(C.GEMM_NN, g, w, split_k=0))
    assert H.close(db.cpu().numpy(), g.cpu().double().sum(0).float().numpy(), 1e-5, 1e-5)
    slot = torch.ones(n_out, device="cuda")
    C.linear_bwd(g, x, w, True, bias=True, db=slot, accumulate_db=True)
    assert H.close(slot.cpu().numpy(), 1.0 + g.cpu().double().sum(0).float().numpy(), 1e-5, 1e-5)
    # and against fp32 torch (loose: different summation order)
    assert H.rel_err(dW.cpu().numpy(), (g.cpu().double().T @ x.cpu().double()).float().numpy()) < 1e-5


@pytest.mark.parametrize("B,n_out,n_in", [(1024, 768, 512), (1024, 691, 768), (1000, 460, 512), (2048, 512, 256), (2047, 333, 385),
                                          (3000, 768, 512), (4095, 200, 300), (8192, 256, 128)])
def test_linear_bwd_balanced_kernel_against_float64_and_itself(C, B, n_out, n_in):
    """gemm_mid_sk_kernel (one 64x64 tile per workgroup at B ~ 1024, even k-step ranges with last-arriver hand-over above): dW with
    accumulation, dX with and without the activation-derivative epilogue, db -- against float64, ragged shapes included -- and
    bit-identical from launch to launch (the hand-over adds partial tiles in ascending k order whoever arrives last)"""
    assert C.workspace_bytes(C.WS_LINEAR_BWD_ZEROED, B, n_out, n_in, 1) > 0, "not a shape of the balanced kernel"
    g = dev(fill.gauss((B, n_out), 74))
    x = dev(fill.gauss((B, n_in), 75))
    w = dev(fill.gauss((n_out, n_in), 76))
    pre = dev(fill.gauss((B, n_in), 77))
    gd, xd, wd, pd = g.cpu().double(), x.cpu().double(), w.cpu().double(), pre.cpu().double()
    for epi, aux in ((C.EPI_NONE, None), (C.EPI_DSILU, pre)):
        dW0, db0 = torch.ones(n_out, n_in, device="cuda"), torch.ones(n_out, device="cuda")
        dW, dX, db = C.linear_bwd(g, x, w, True, epi, aux, dW=dW0, accumulate=True, bias=True, db=db0, accumulate_db=True)
        want_dX = gd @ wd
        if aux is not None:
            sg = torch.sigmoid(pd)
            want_dX = want_dX * (sg * (1 + pd * (1 - sg)))
        assert H.rel_err(dW.cpu().numpy(), (1.0 + gd.T @ xd).float().numpy()) < 2e-6
        assert H.rel_err(dX.cpu().numpy(), want_dX.float().numpy()) < 2e-6
        assert H.close(db.cpu().numpy(), (1.0 + gd.sum(0)).float().numpy(), 1e-5, 1e-5)
    first = C.linear_bwd(g, x, w, True, bias=True)
    for _ in range(5):
        again = C.linear_bwd(g, x, w, True, bias=True)
        assert all(torch.equal(a, b) for a, b in zip(first, again))
    only, none = C.linear_bwd(g, x, w, False)
    assert none is None and H.rel_err(only.cpu().numpy(), (gd.T @ xd).float().numpy()) < 2e-6
    # the arrival counters at the head of the lane's workspace are zero again
    lane = C._lane_ws(g.device, 4)
    assert int(lane[:4096].view(torch.int32).abs().sum()) == 0
    # every stream owns its counters (two Linear backwards on different streams must never share them), and an error path re-zeroes them
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        other = C._lane_ws(g.device, 4)
    assert other.data_ptr() != lane.data_ptr() and C.lane_counters_clean()
    lane[3] = 1.0  # what a launch that died mid-way would leave behind
    assert not C.lane_counters_clean()
    with pytest.raises(RuntimeError):
        C.linear_bwd(g, x, w, True, C.EPI_DSILU, None)  # refused by the library (a D* epilogue without aux): reported, workspaces reset
    assert C.lane_counters_clean()


@pytest.mark.parametrize("B,n_out,n_in", [(1024, 768, 512), (1024, 691, 768), (1000, 460, 512), (1024, 345, 691), (2048, 512, 256)])
def test_linear_bwd_on_an_announced_side_stream_takes_the_small_workgroups(C, B, n_out, n_in):
    """co_resident: on a stream announced with register_ws_lane the balanced kernel runs eight-wave workgroups (two per CU) where it
    can take whole tiles -- same contract: float64 parity incl. the gated epilogue, accumulate, db; launch-to-launch bit identity;
    clean counters; within rounding of the sixteen-wave form"""
    g, x, w = dev(fill.gauss((B, n_out), 84)), dev(fill.gauss((B, n_in), 85)), dev(fill.gauss((n_out, n_in), 86))
    y = dev(np.maximum(fill.gauss((B, n_in), 87), 0.0))
    gd, xd, wd = g.cpu().double(), x.cpu().double(), w.cpu().double()
    main_form = C.linear_bwd(g, x, w, True, C.EPI_DRELU, y, bias=True, dx_scale=1.25)
    side = torch.cuda.Stream()
    C.register_ws_lane(side)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dW0, db0 = torch.ones(n_out, n_in, device="cuda"), torch.ones(n_out, device="cuda")
        dW, dX, db = C.linear_bwd(g, x, w, True, C.EPI_DRELU, y, dW=dW0, accumulate=True, bias=True, db=db0, accumulate_db=True, dx_scale=1.25)
        first = C.linear_bwd(g, x, w, True, C.EPI_DRELU, y, bias=True, dx_scale=1.25)
        for _ in range(4):
            again = C.linear_bwd(g, x, w, True, C.EPI_DRELU, y, bias=True, dx_scale=1.25)
            assert all(torch.equal(a, b) for a, b in zip(first, again))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want_dX = (gd @ wd) * (y.cpu().double() > 0) * 1.25
    assert H.rel_err(dW.cpu().numpy(), (1.0 + gd.T @ xd).float().numpy()) < 2e-6
    assert H.rel_err(dX.cpu().numpy(), want_dX.float().numpy()) < 2e-6
    assert H.close(db.cpu().numpy(), (1.0 + gd.sum(0)).float().numpy(), 1e-5, 1e-5)
    for a, b in zip(first, main_form):
        assert H.rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-6
    assert C.lane_counters_clean()


@pytest.mark.parametrize("B,mode,norm", [(1024, 3, True), (50, 3, True), (16, 2, False), (333, 3, False)])
def test_bottleneck_launch_is_bit_identical_to_the_separate_launches(C, B, mode, norm):
    """enc(256->128->32) + 3x256 RQ + dec(32->128->256) in one launch vs gemm / rq_forward one by one"""
    K2, N2, Nd0, Nd1, L, K = 256, 128, 128, 256, 3, 256
    assert C.bottleneck_eligible(B, K2, N2, Nd0, Nd1, L, K)
    h1 = dev(fill.uniform((B, K2), 90, -1, 1))
    W2, W3 = dev(fill.uniform((N2, K2), 91, -0.08, 0.08)), dev(fill.uniform((32, N2), 92, -0.1, 0.1))
    Wd0, Wd1 = dev(fill.uniform((Nd0, 32), 93, -0.2, 0.2)), dev(fill.uniform((Nd1, Nd0), 94, -0.1, 0.1))
    _, tables = _rq_inputs(B, L, K, 95)
    cb, cc = C.codebook_prepare([dev(t) for t in tables], [norm and i == 0 for i in range(L)])
    o = C.bottleneck_fwd(h1, W2, W3, cb, cc, norm, mode, 0.4, Wd0, Wd1)
    pre2 = torch.empty(B, N2, device="cuda")
    h2 = C.gemm(C.GEMM_NT, h1, W2, epilogue=C.EPI_SILU, aux=pre2)
    y = C.gemm(C.GEMM_NT, h2, W3)
    z, ids, emb_cat, emb_sum, _, qloss = C.rq_forward(y, cb, cc, norm, mode, True, 0.4)
    pre_d0 = torch.empty(B, Nd0, device="cuda")
    d0 = C.gemm(C.GEMM_NT, emb_sum, Wd0, epilogue=C.EPI_SILU, aux=pre_d0)
    pre_d1 = torch.empty(B, Nd1, device="cuda")
    d1 = C.gemm(C.GEMM_NT, d0, Wd1, epilogue=C.EPI_SILU, aux=pre_d1)
    want = dict(pre2=pre2, h2=h2, y=y, z=z, ids=ids, emb_cat=emb_cat, emb_sum=emb_sum, qloss=qloss, pre_d0=pre_d0, d0=d0, pre_d1=pre_d1, d1=d1)
    for k, v in want.items():
        assert torch.equal(o[k], v), k
    # the same launch carrying the debug statistics (id census + |emb_out| per level) against hidvae_id_stats; repeated calls walk
    # the census table's generations, and a batch of copies has few distinct tuples
    table = C.census_scratch(B, "cuda")  # one caller-owned table for all the calls below
    for rep in range(3):
        oc = C.bottleneck_fwd(h1, W2, W3, cb, cc, norm, mode, 0.4, Wd0, Wd1, id_stats=True, scratch=table)
        norms, pu = C.id_stats(oc["emb_cat"], oc["ids"])
        for k, v in want.items():
            assert torch.equal(oc[k], v), k
        assert torch.equal(oc["embs_norm"], norms) and float(oc["p_unique"]) == float(pu), rep
    hdup = h1[torch.arange(B, device="cuda") % 5].contiguous()
    od = C.bottleneck_fwd(hdup, W2, W3, cb, cc, norm, mode, 0.4, Wd0, Wd1, id_stats=True, scratch=table)
    distinct = len({tuple(r) for r in od["ids"].cpu().tolist()})
    assert float(od["p_unique"]) == float(torch.tensor(distinct, dtype=torch.float32) / torch.tensor(B, dtype=torch.float32))
    assert torch.equal(od["embs_norm"], C.id_stats(od["emb_cat"], od["ids"])[0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,D", [(7, 20, 64), (33, 5, 128), (4, 50, 6), (1, 1, 8)])
def test_padded_to_jagged_and_its_backward(C, dtype, B, N, D):
    """the stage-2 jagged copy (reference ops/triton/jagged.py) against its definition: values = the first lengths[b] rows of
    every entry back to back; gradient = the same rows scattered back, zeros on the padding"""
    from hidvae_amd.ops_hip.jagged import jagged_to_flattened_tensor, padded_to_jagged_tensor
    g = torch.Generator().manual_seed(B * 100 + N)
    x = torch.randn(B, N, D, generator=g).to(dtype).cuda().requires_grad_(True)
    lengths = torch.randint(0, N + 1, (B,), generator=g)
    lengths[0] = N
    nt = padded_to_jagged_tensor(x, lengths.cuda(), N)
    vals = jagged_to_flattened_tensor(nt)
    want = torch.cat([x.detach()[b, :int(lengths[b])] for b in range(B)], 0)
    assert vals.shape == want.shape and torch.equal(vals.detach(), want)
    assert nt.is_nested and [int(t.shape[0]) for t in nt.unbind()] == lengths.tolist()
    w = torch.randn(want.shape, generator=g).to(dtype).cuda()
    (vals * w).sum().backward()
    gx = torch.zeros(B, N, D, dtype=dtype, device="cuda")
    row = 0
    for b in range(B):
        gx[b, :int(lengths[b])] = w[row:row + int(lengths[b])]
        row += int(lengths[b])
    assert torch.equal(x.grad, gx)


@pytest.mark.parametrize("M,N,relu,masked", [(1024, 768, True, True), (1000, 691, False, True), (37, 230, True, False), (2048, 256, False, False),
                                             (5, 1024, True, True)])
def test_layernorm_forward_and_backward_against_float64(C, M, N, relu, masked):
    """hidvae_layernorm_fwd + hidvae_layernorm_bwd_all (gx and the affine gradients from one pass over the rows + a fixed-order finish)
    against torch float64 autograd of drop(relu(LN(x))) + res; gradient slots with accumulation; launch-to-launch bit identity"""
    x = dev(fill.gauss((M, N), 80))
    gamma, beta = dev(fill.uniform((N,), 81, 0.5, 1.5)), dev(fill.uniform((N,), 82, -0.3, 0.3))
    gy = dev(fill.gauss((M, N), 83))
    mask = dev((fill.uniform((M, N), 84, 0, 1) < 0.7).astype(np.float32)) if masked else None
    scale = 1.0 / 0.7 if masked else 1.0
    y, mean, rstd = C.layernorm_fwd(x, gamma, beta, 1e-5, relu, mask, scale, None)
    xd = x.cpu().double().requires_grad_(True)
    gd, bd = gamma.cpu().double().requires_grad_(True), beta.cpu().double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (N,), gd, bd, 1e-5)
    if relu:
        ref = torch.relu(ref)
    if masked:
        ref = ref * (mask.cpu().double() * scale)
    assert H.close(y.cpu().numpy(), ref.detach().float().numpy(), 2e-5, 2e-5)
    ref.backward(gy.cpu().double())
    gg0, gb0 = torch.ones(N, device="cuda"), torch.ones(N, device="cuda")
    gx, gg, gb = C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, scale, gg=gg0, gb=gb0, accumulate=True)
    # (an element whose pre-activation is within rounding of 0 may sit on the other side of the ReLU in float64: bulk-tight)
    assert H.rel_err(gx.cpu().numpy(), xd.grad.float().numpy()) < 3e-5
    assert H.rel_err(gg.cpu().numpy(), (1.0 + gd.grad).float().numpy()) < 3e-5
    assert H.rel_err(gb.cpu().numpy(), (1.0 + bd.grad).float().numpy()) < 3e-5
    a = C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, scale)
    b = C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, scale)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    none, gg2, gb2 = C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, relu, mask, scale, need_gx=False)
    assert none is None and torch.equal(gg2, a[1]) and torch.equal(gb2, a[2])


def test_linear_bwd_hand_over_is_stable_under_traffic(C):
    """equal-range schedule of the balanced kernel: partial tiles cross workgroups (and XCDs) through cache-bypassing stores / loads and
    an arrival counter -- 600 launches with a large copy running on a second stream, every result bit-identical to the first
    (the long version, 20,000 launches on five shapes: scratch/sk_stress.py)"""
    B, n_out, n_in = 2048, 768, 512
    g, x, w = dev(fill.gauss((B, n_out), 90)), dev(fill.gauss((B, n_in), 91)), dev(fill.gauss((n_out, n_in), 92))
    ref = [t.clone() for t in C.linear_bwd(g, x, w, True, bias=True)]
    side = torch.cuda.Stream()
    big_a, big_b = torch.randn(32 << 20, device="cuda"), torch.empty(32 << 20, device="cuda")
    for it in range(600):
        if it % 40 == 0:
            with torch.cuda.stream(side):
                big_b.copy_(big_a)
        out = C.linear_bwd(g, x, w, True, bias=True)
        if it % 20 == 0 or it >= 590:
            assert all(torch.equal(a, b) for a, b in zip(out, ref)), f"launch {it} differs"
    torch.cuda.synchronize()


def test_gather_rows_is_index_select_on_every_table(C):
    """hidvae_gather_rows (reference data/tags_processed.py:112-150: the batch's ids index item_data, tags_emb and tags_indices): one
    launch, every table, bit-exact against torch.index_select; ragged row sizes (16-byte and dword paths), repeated ids, one row, a
    destination that is a training step's input buffer (ResidentItemData.gather_into), and the refusals"""
    import types
    from hidvae_amd.data.items import RandomBatches, ResidentItemData
    g = torch.Generator().manual_seed(11)
    for n, B, W in ((50, 16, 8), (1000, 1024, 768), (7, 1, 5), (300, 257, 33)):
        x = torch.randn(n, W, generator=g).cuda()
        te = torch.randn(n, 3, W, generator=g).cuda()
        ti = torch.randint(-1, 40, (n, 3), generator=g).cuda()
        idx = torch.randint(0, n, (B,), generator=g).cuda()
        outs = [torch.full((B, W), -7.0, device="cuda"), torch.full((B, 3, W), -7.0, device="cuda"),
                torch.full((B, 3), -7, dtype=torch.int64, device="cuda")]
        C.gather_rows(idx, [x, te, ti], outs)
        for t, o in zip((x, te, ti), outs):
            assert torch.equal(o, torch.index_select(t, 0, idx))
        one = torch.empty(B, W, device="cuda")
        C.gather_rows(idx, [x], [one])
        assert torch.equal(one, x[idx])
    # the loader's path: the same batches as the indexing form, written into given buffers
    data = ResidentItemData(x, te, ti)
    a, b = RandomBatches(data, 64, seed=5), RandomBatches(data, 64, seed=5)
    out = types.SimpleNamespace(x=torch.empty(64, W, device="cuda"), tags_emb=torch.empty(64, 3, W, device="cuda"),
                                tags_indices=torch.empty(64, 3, dtype=torch.long, device="cuda"))
    for _ in range(6):  # crosses a re-permutation
        fresh, given = a.next(), b.next(out=out)
        assert given is out
        assert torch.equal(fresh.x, out.x) and torch.equal(fresh.tags_emb, out.tags_emb) and torch.equal(fresh.tags_indices, out.tags_indices)
    with pytest.raises(RuntimeError):
        C.gather_rows(idx.int(), [x], [one])
    with pytest.raises(RuntimeError):
        C.gather_rows(idx, [x], [torch.empty(B, W + 1, device="cuda")])
    with pytest.raises(RuntimeError):
        C.gather_rows(idx, [x.half()], [torch.empty(B, W, dtype=torch.half, device="cuda")])  # rows of 66 bytes: not whole dwords
