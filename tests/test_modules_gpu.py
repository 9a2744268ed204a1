"""GPU: the reference's small modules as stand-alone callables on the HIP path -- ReconstructionLoss / QuantizeLoss
(reference modules/loss.py:7-12, 36-44), SemanticIdUniquenessLoss called directly (h_rqvae.py:41-105), distributions/gumbel.py --
each against the torch expression the reference evaluates, values and gradients."""
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
