"""GPU: size-independent properties of the hot path at BASELINE.json's full sizes (where a complete oracle run per test
would take too long to repeat per case): configs[3] B=8192 3x256 and configs[4] B=4096 4x1024, plus a corpus-sized launch."""
import numpy as np
import pytest
import torch

from oracle import exact, fill

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def C():
    import hidvae_amd  # noqa: F401
    from hidvae_amd import _C
    _C.lib()
    return _C


def tables(L, K, seed):
    return [torch.from_numpy(fill.uniform((K, 32), seed + i, -1, 1) * np.float32(1.0 if i == 0 else 0.35 * 0.5 ** i)).cuda() for i in range(L)]


@pytest.mark.parametrize("B,L,K", [(8192, 3, 256), (4096, 4, 1024)])
def test_rq_at_baseline_sizes(C, B, L, K):
    tabs = tables(L, K, 60)
    y = torch.from_numpy(fill.gauss((B, 32), 61)).cuda()
    flags = [i == 0 for i in range(L)]
    cb, cc = C.codebook_prepare(tabs, flags)
    z, ids, emb_cat, emb_sum, res, qloss = C.rq_forward(y, cb, cc, True, 3, True, 0.4, want_res=True)
    # (1) bit-exact against the C oracle at the full size (ids AND floats)
    want = exact.rq_forward(y.cpu().numpy(), [t.cpu().numpy() for t in tabs], True, True, 3, True, 0.4)
    assert np.array_equal(ids.cpu().numpy(), want["ids"])
    assert np.array_equal(emb_cat.cpu().numpy(), want["emb_cat"]) and np.array_equal(qloss.cpu().numpy(), want["loss"])
    # (2) determinism: a second launch is bit-identical
    again = C.rq_forward(y, cb, cc, True, 3, True, 0.4, want_res=True)
    for a, b in zip((z, ids, emb_cat, emb_sum, res, qloss), again):
        assert torch.equal(a, b)
    # (3) permutation equivariance: items are independent, so permuting the batch permutes every output bit for bit
    perm = torch.from_numpy(fill.perm(B, 62)).cuda()
    zp, idp, ecp, esp, _, qlp = C.rq_forward(y[perm].contiguous(), cb, cc, True, 3, True, 0.4)
    assert torch.equal(idp, ids[perm]) and torch.equal(ecp, emb_cat[perm]) and torch.equal(qlp, qloss[perm])
    # (4) residual chain: res_{i+1} = res_i - o_i and sum_i o_i = emb_sum (exactly the kernel's own order)
    r = res.view(B, L, 32)
    o = emb_cat.view(B, L, 32)
    for i in range(L - 1):
        assert torch.equal(r[:, i + 1], r[:, i] - o[:, i])
    acc = o[:, 0].clone()
    for i in range(1, L):
        acc = acc + o[:, i]
    assert torch.equal(acc, emb_sum)
    # (5) rotation trick keeps the norm of its input: |o_i| = |r_i| (quantize.py:34-45), and ids are valid codes
    assert float(((o.norm(dim=-1) - r.norm(dim=-1)).abs() / r.norm(dim=-1)).max()) < 1e-5
    assert int(ids.min()) >= 0 and int(ids.max()) < K
    # (6) eval branch: same ids at level 0, o = selected code, and a code looked up in its own codebook returns itself
    _, ide, ece, _, rese, qle = C.rq_forward(y, cb, cc, True, 3, False, 0.4, want_res=True)
    assert torch.equal(ide[:, 0], ids[:, 0])
    assert torch.equal(ece.view(B, L, 32)[:, 0], cb[0][ide[:, 0]])
    probe = cb[0][:K].contiguous()
    _, idc, _, _, _, qlc = C.rq_forward(probe, cb[:1].contiguous(), cc[:1].contiguous(), False, 2, False, 0.4)
    assert torch.equal(idc[:, 0].cpu(), torch.arange(K)) and float(qlc.abs().max()) < 1e-10


def test_gemm_linearity_and_tiling_independence(C):
    """(aX1 + bX2) W^T == a X1 W^T + b X2 W^T within fp32 rounding at the step's largest shape; the exact kernel gives the
    same bits whatever batch the rows arrive in (row independence)."""
    B = 8192
    x1 = torch.from_numpy(fill.gauss((B, 768), 70)).cuda()
    x2 = torch.from_numpy(fill.gauss((B, 768), 71)).cuda()
    w = torch.from_numpy(fill.uniform((512, 768), 72, -0.04, 0.04)).cuda()
    lhs = C.gemm(C.GEMM_NT, (0.5 * x1 + 2.0 * x2).contiguous(), w)
    rhs = 0.5 * C.gemm(C.GEMM_NT, x1, w) + 2.0 * C.gemm(C.GEMM_NT, x2, w)
    assert float((lhs - rhs).abs().max() / rhs.abs().max()) < 2e-6
    full = C.gemm(C.GEMM_NT, x1, w)
    part = C.gemm(C.GEMM_NT, x1[1000:1777].contiguous(), w)
    assert torch.equal(full[1000:1777], part)


def test_corpus_sized_launch_against_oracle_sample(C):
    """1,048,576 items through the fused VQ: every 997th item re-checked bit for bit by the C oracle, and the id histogram is
    consistent (a checksum of per-level id sums recomputed from the outputs)."""
    N, L, K = 1 << 20, 3, 256
    tabs = tables(L, K, 80)
    y = torch.randn(N, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    cb, cc = C.codebook_prepare(tabs, [True, False, False])
    _, ids, emb_cat, _, _, qloss = C.rq_forward(y, cb, cc, True, 3, False, 0.4)
    pick = torch.arange(0, N, 997, device="cuda")
    want = exact.rq_forward(y[pick].cpu().numpy(), [t.cpu().numpy() for t in tabs], True, True, 3, False, 0.4)
    assert np.array_equal(ids[pick].cpu().numpy(), want["ids"])
    assert np.array_equal(qloss[pick].cpu().numpy(), want["loss"])
    # eval outputs are codebook rows: gathering them by id must reproduce emb_cat exactly
    for i in range(L):
        assert torch.equal(emb_cat[:, i * 32:(i + 1) * 32], cb[i][ids[:, i]])
    # the training (rotation-trick) form of the same large-batch kernel variant (16 waves per workgroup), sampled likewise
    M = 1 << 17
    z, ids, emb_cat, emb_sum, _, qloss = C.rq_forward(y[:M], cb, cc, True, 3, True, 0.4)
    pick = torch.arange(0, M, 331, device="cuda")
    want = exact.rq_forward(y[pick].cpu().numpy(), [t.cpu().numpy() for t in tabs], True, True, 3, True, 0.4)
    for got, key in ((ids, "ids"), (z, "z"), (emb_cat, "emb_cat"), (emb_sum, "emb_sum"), (qloss, "loss")):
        assert np.array_equal(got[pick].cpu().numpy(), want[key]), key


def test_training_step_is_bitwise_reproducible():
    """Two identical steps from identical state give identical losses and gradients (fixed-order reductions, no atomics)."""
    from oracle import torch_oracle as O
    from tests.test_model_gpu import build_model, make_batch
    from hidvae_amd.rand import InjectedRand
    cfg = O.Cfg(tag_class_counts=[38, 168, 348], use_focal_loss=True, focal_loss_params={"gamma_0": 2.7, "alpha_0": 0.24},
                commitment_weight=0.4)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    x, te, ti = O.formula_batch(cfg, 2048, seed=3, tagged=True)
    outs = []
    for _ in range(2):
        m = build_model(cfg, P).train()
        m.rand = InjectedRand(O.FormulaRand())
        out = m(make_batch(x, te, ti), gumbel_t=0.2)
        out.loss.backward()
        torch.cuda.synchronize()
        outs.append((out.loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def _philox4x32_10(key, c):
    """reference Philox4x32-10 (Salmon et al., SC'11) on numpy uint64 lanes: key (k0, k1), counters c [n, 4] -> words [n, 4]"""
    c = [c[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, 1)


def test_device_rand_provider_properties(C):
    """The production randomness provider (no golden can pin a device RNG stream).  Dropout is decided INSIDE the producing launches
    by a counter-based generator: the generator is Philox4x32-10 bit for bit (known-answer vector of the Random123 distribution +
    a numpy restatement on this provider's counters); keep rates are the requested ones; sites, steps and seeds give uncorrelated
    masks; begin_step advances the device step; a launch given a DropSpec computes exactly what it computes given the materialised
    mask (LayerNorm, BatchNorm, GEMM epilogue), and the backward -- which never sees the mask -- equals the mask-based backward.
    Mixup: the pairing is a permutation of the valid rows among themselves with a consistent inverse, lam lies in (0,1)."""
    from hidvae_amd.rand import DeviceRand
    dev = torch.device("cuda")
    assert _philox4x32_10((0, 0), np.zeros((1, 4), np.uint64))[0].tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = DeviceRand(0.2, seed=0x1234567855AA33CC)
    r.begin_step(dev)
    r.begin_step(dev)
    assert r.state(dev).tolist() == [0x1234567855AA33CC, 2]
    spec = r.dropout_keep((300, 257), 0.4, dev)
    spec2 = r.dropout_keep((300, 257), 0.4, dev)
    assert (spec.site, spec2.site) == (0, 1)
    m = C.dropout_mask(spec, (300, 257)).cpu().numpy()
    n = 300 * 257
    ctr = np.stack([np.arange(n, dtype=np.uint64), np.zeros(n, np.uint64), np.full(n, spec.site, np.uint64), np.full(n, 2, np.uint64)], 1)
    words = _philox4x32_10((0x55AA33CC, 0x12345678), ctr)[:, 0]
    want = (words >= np.uint64(spec.threshold)).astype(np.float32).reshape(300, 257)
    assert np.array_equal(m, want), "the device generator is not Philox4x32-10 on (element, site, step)"
    for p in (0.1, 0.4, 0.55):
        mk = C.dropout_mask(r.dropout_keep((1024, 768), p, dev), (1024, 768))
        assert set(torch.unique(mk).cpu().tolist()) <= {0.0, 1.0} and abs(float(mk.mean()) - (1 - p)) < 3e-3
    a = C.dropout_mask(spec, (300, 257))
    b = C.dropout_mask(spec2, (300, 257))                      # another site
    r.begin_step(dev)
    c = C.dropout_mask(C.DropSpec(r.state(dev), spec.site, 0.4), (300, 257))  # the same site, next step
    d = C.dropout_mask(DeviceRand(0.2, seed=99).dropout_keep((300, 257), 0.4, dev), (300, 257))  # another seed
    corr = lambda u, v: abs(float(torch.corrcoef(torch.stack([u.flatten(), v.flatten()]))[0, 1]))
    assert corr(a, b) < 0.02 and corr(a, c) < 0.02 and corr(a, d) < 0.02 and not torch.equal(a, c)
    # ---- launches: DropSpec == materialised mask, forward and backward
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(256, 691, device="cuda", generator=g)
    gamma, beta = torch.randn(691, device="cuda", generator=g), torch.randn(691, device="cuda", generator=g)
    sp = r.dropout_keep(x.shape, 0.4, dev)
    mask = C.dropout_mask(sp, x.shape)
    sc = 1.0 / 0.6
    y0, mean, rstd = C.layernorm_fwd(x, gamma, beta, 1e-5, True, mask, sc, None)
    y1, _, _ = C.layernorm_fwd(x, gamma, beta, 1e-5, True, sp, sc, None)
    assert torch.equal(y0, y1)
    gy = torch.randn_like(x)
    gx0, _, _ = C.layernorm_bwd_all(gy, x, gamma, beta, mean, rstd, True, mask, sc)
    gx1, _ = C.layernorm_bwd_partial(gy, x, gamma, beta, mean, rstd, True, y1, sc)
    assert torch.equal(gx0, gx1)
    rm, rv = torch.zeros(691, device="cuda"), torch.ones(691, device="cuda")
    b0, sm, sr = C.batchnorm_fwd(x, gamma, beta, 1e-5, 0.1, True, rm.clone(), rv.clone(), True, mask, sc)
    b1, _, _ = C.batchnorm_fwd(x, gamma, beta, 1e-5, 0.1, True, rm.clone(), rv.clone(), True, sp, sc)
    assert torch.equal(b0, b1)
    h0 = C.batchnorm_bwd(gy, x, gamma, beta, sm, sr, True, mask, sc)
    h1 = C.batchnorm_bwd(gy, x, gamma, beta, sm, sr, True, None, sc, y_out=b1)
    assert all(torch.equal(u, v) for u, v in zip(h0, h1))
    for (Mg, Ng, Kg) in ((256, 691, 768), (1000, 345, 691), (64, 38, 115), (4096, 768, 512)):
        A, W, bias = torch.randn(Mg, Kg, device="cuda", generator=g), torch.randn(Ng, Kg, device="cuda", generator=g), torch.randn(Ng, device="cuda", generator=g)
        sg = r.dropout_keep((Mg, Ng), 0.2, dev)
        mg = C.dropout_mask(sg, (Mg, Ng))
        o0 = C.gemm(C.GEMM_NT, A, W, bias=bias, epilogue=C.EPI_RELU, mask=mg, mask_scale=1.25, split_k=0)
        o1 = C.gemm(C.GEMM_NT, A, W, bias=bias, epilogue=C.EPI_RELU, mask=sg, mask_scale=1.25, split_k=0)
        assert torch.equal(o0, o1), (Mg, Ng, Kg)
    # ---- mixup
    gt = torch.Generator().manual_seed(5)
    tg = torch.randint(0, 7, (500, 3), generator=gt)
    tg[torch.rand(500, generator=gt) < 0.3, 1] = -1
    tg[:, 2] = -1
    tg[7, 2] = 3  # a level with a single valid row
    tgd = tg.to(dev)
    plans = r.mixup_all(tgd, dev)
    for level, (partner, inverse, lam) in enumerate(plans):
        p, inv, t = partner.cpu(), inverse.cpu(), tg[:, level]
        valid = t >= 0
        assert (p[~valid] == -1).all() and (inv[~valid] == -1).all()
        assert sorted(p[valid].tolist()) == sorted(torch.nonzero(valid).flatten().tolist())  # a permutation of the valid rows
        assert (inv[p[valid]] == torch.nonzero(valid).flatten()).all()
        assert 0.0 <= float(lam) <= 1.0
    r.begin_step(dev)
    again = r.mixup_all(tgd, dev)
    assert not torch.equal(again[0][0], plans[0][0]), "the pairing did not change with the step"
    r.prepare_mixup(tgd, dev)
    assert r.mixup_partner(tgd[:, 1], dev, level=1)[0].shape == (500,)
    single = r.mixup_partner(tgd[:, 0], dev)
    assert sorted(single[0].cpu().tolist()) == list(range(500))


@pytest.mark.parametrize("mode,training", [(3, True), (2, True), (3, False)])
@pytest.mark.parametrize("kind", ["spread", "near_duplicates", "exact_duplicates", "tiny_residuals"])
@pytest.mark.parametrize("N,K", [(70000, 256), (270001, 256), (262144 + 77, 96)])
def test_prefilter_kernel_is_bit_identical_to_the_exact_kernel(C, mode, training, kind, N, K):
    """Batches >= 65536 take the split-bf16 prefilter kernel (approximate scores on bf16 MFMA, exact re-search where the two best
    scores are closer than the error bound), batches >= 262144 its 32-items-per-wave form (32x32x16 MFMA, index packed into the
    scores' low bits; K = 96 takes the not-unrolled instance); smaller launches take the exact fp32 kernel.  Same items through
    both: every output must match bit for bit -- also when most items are ambiguous (near-duplicate / duplicate codes) and when
    residual norms collapse (levels whose codes are far larger than the residual)."""
    L = 3
    g = torch.Generator(device="cuda").manual_seed(17)
    tabs = tables(L, K, 81)
    if kind == "near_duplicates":
        for t in tabs:
            t[K // 2:] = t[:K // 2] * (1.0 + 1e-6 * torch.randn(K // 2, 1, device="cuda", generator=g))
    elif kind == "exact_duplicates":
        for t in tabs:
            t[K // 2:] = t[:K // 2]
    elif kind == "tiny_residuals":
        tabs[1] = tabs[1] * 50.0
        tabs[2] = tabs[2] * 1e-4
    y = torch.randn(N, 32, device="cuda", generator=g)
    cb, cc = C.codebook_prepare(tabs, [True, False, False])
    big = C.rq_forward(y, cb, cc, True, mode, training, 0.4, want_res=True)
    cuts = list(range(0, N, 60000)) + [N]  # every piece below 65536 items: the exact kernel
    parts = [C.rq_forward(y[lo:hi].contiguous(), cb, cc, True, mode, training, 0.4, want_res=True) for lo, hi in zip(cuts[:-1], cuts[1:])]
    names = ("z", "ids", "emb_cat", "emb_sum", "res_cat", "qloss")
    for j, name in enumerate(names):
        want = torch.cat([p[j] for p in parts], 0)
        assert torch.equal(big[j], want), (kind, name, int((big[j] != want).sum()))


@pytest.mark.parametrize("kind", ["spread", "near_duplicates", "exact_duplicates", "tiny_residuals"])
@pytest.mark.parametrize("N,K", [(70000, 256), (270001, 256), (1000, 256), (66000, 96)])
def test_ids_only_launch_gives_the_ids_of_the_full_eval_launch(C, kind, N, K):
    """hidvae_rq_forward with every output but `ids` NULL (the tokenizer's corpus pass) takes, from 65536 items and K = 256 on, the
    ids-only form of the prefilter kernel: no output rows, no winner fetch after the last level, 16 waves per workgroup.  Its ids
    must be those of the full eval-mode launch (pieces below 65536 items: the exact fp32 kernel) bit for bit, ambiguous items
    (duplicate / near-duplicate codes) and collapsing residual norms included; other sizes take the ordinary kernels with NULL
    outputs."""
    L = 3
    g = torch.Generator(device="cuda").manual_seed(19)
    tabs = tables(L, K, 83)
    if kind == "near_duplicates":
        for t in tabs:
            t[K // 2:] = t[:K // 2] * (1.0 + 1e-6 * torch.randn(K // 2, 1, device="cuda", generator=g))
    elif kind == "exact_duplicates":
        for t in tabs:
            t[K // 2:] = t[:K // 2]
    elif kind == "tiny_residuals":
        tabs[1] = tabs[1] * 50.0
        tabs[2] = tabs[2] * 1e-4
    y = torch.randn(N, 32, device="cuda", generator=g)
    cb, cc = C.codebook_prepare(tabs, [True, False, False])
    for normalize in (True, False):
        got = C.rq_ids(y, cb, cc, normalize)
        cuts = list(range(0, N, 60000)) + [N]
        want = torch.cat([C.rq_forward(y[lo:hi].contiguous(), cb, cc, normalize, 2, False, 0.4)[1] for lo, hi in zip(cuts[:-1], cuts[1:])], 0)
        assert torch.equal(got, want), (kind, normalize, int((got != want).sum()))


def test_mixup_plan_kernel_properties(C):
    """hidvae_mixup_plan (one launch for the pairings and lambdas of all levels): partner is a permutation of the valid rows among
    themselves with the matching inverse, -1 on invalid rows, for ragged / maximal / single-row shapes; over many draws lam has the
    mean and variance of Beta(alpha, alpha) and the pairing is not biased towards any row."""
    g = torch.Generator(device="cuda").manual_seed(23)
    for B, L in ((1, 1), (7, 3), (1000, 3), (1025, 2), (4096, 8), (8192, 3), (16384, 2), (12345, 3)):
        t = torch.randint(0, 5, (B, L), device="cuda", generator=g)
        t[torch.rand(B, device="cuda", generator=g) < 0.3, 0] = -1
        if L > 1:
            t[:, L - 1] = -1
            t[B // 2, L - 1] = 2  # a level with one valid row
        u = torch.rand(L, B + 64, device="cuda", generator=g)
        partner, inverse, lam = C.mixup_plan(t, u, 0.2)
        for l in range(L):
            valid = (t[:, l] >= 0).cpu()
            p, inv = partner[l].cpu(), inverse[l].cpu()
            rows = torch.nonzero(valid).flatten()
            assert (p[~valid] == -1).all() and (inv[~valid] == -1).all()
            assert sorted(p[valid].tolist()) == rows.tolist()
            assert (inv[p[valid]] == rows).all()
            assert 0.0 <= float(lam[l]) <= 1.0
        again = C.mixup_plan(t, u, 0.2)  # same uniforms, same plan
        assert torch.equal(again[0], partner) and torch.equal(again[1], inverse) and torch.equal(again[2], lam)
        if B > 1:  # the pairing is the valid rows sorted by (key, row): the definition, whatever the size
            for l in range(L):
                rows = torch.nonzero(t[:, l] >= 0).flatten()
                if len(rows):
                    order = rows[torch.argsort(u[l, rows] * 0 + u[l, rows], stable=True)]
                    assert torch.equal(partner[l, rows], order)
    B, L, alpha = 64, 8, 0.2
    t = torch.zeros(B, L, dtype=torch.int64, device="cuda")
    lams, firsts = [], []
    for _ in range(250):
        partner, _, lam = C.mixup_plan(t, torch.rand(L, B + 64, device="cuda", generator=g), alpha)
        lams.append(lam)
        firsts.append(partner[:, 0])
    lams, firsts = torch.cat(lams).double().cpu(), torch.cat(firsts).cpu()
    assert abs(float(lams.mean()) - 0.5) < 0.03
    assert abs(float(lams.var()) - 1.0 / (4 * (2 * alpha + 1))) < 0.02   # Var Beta(a,a) = 1 / (4 (2a + 1))
    counts = torch.bincount(firsts, minlength=B).double()  # partner of row 0: uniform over the 64 rows (2000 draws)
    assert float(((counts - counts.mean()) ** 2 / counts.mean()).sum()) < 120.0  # chi-square, 63 dof (mean 63, sd 11)
