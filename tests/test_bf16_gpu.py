"""GPU: the OPT-IN bf16 throughput mode of the large-batch Linear layers (hidvae_gemm_bf16; BASELINE config 2 is quoted "bf16").
Not a parity path -- parity is claimed on fp32 -- so these tests BOUND its deviation: the kernel against float64, and a whole
B = 8192 train step against the fp32 step (semantic-id agreement, loss and gradient deviation)."""
import types

import numpy as np
import pytest
import torch

from oracle import fill, torch_oracle as O
from tests import helpers as H
from tests.test_model_gpu import build_model

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("layout,M,N,K", [(0, 8192, 512, 768), (1, 4096, 768, 512), (2, 512, 768, 8192), (0, 1000, 300, 333),
                                          (2, 230, 691, 4096), (1, 4100, 96, 691)])
def test_bf16_gemm_against_float64(layout, M, N, K):
    import hidvae_amd  # noqa: F401
    from hidvae_amd import _C
    shapeA = (K, M) if layout == _C.GEMM_TN else (M, K)
    shapeB = (N, K) if layout == _C.GEMM_NT else (K, N)
    A, B = fill.gauss(shapeA, 11), fill.gauss(shapeB, 12) * 0.05
    bias = fill.gauss((N,), 13)
    a64 = torch.from_numpy(A).double()
    b64 = torch.from_numpy(B).double()
    opA = a64.T if layout == _C.GEMM_TN else a64
    opB = b64.T if layout == _C.GEMM_NT else b64
    ref = opA @ opB + torch.from_numpy(bias).double()
    pre = torch.empty(M, N, device="cuda")
    got = _C.gemm_bf16(layout, dev(A), dev(B), bias=dev(bias), epilogue=_C.EPI_SILU, aux=pre)
    scale = float(ref.abs().max())
    assert float((pre.cpu().double() - ref).abs().max()) <= 1e-2 * scale  # bf16 operands (8 mantissa bits), fp32 accumulation
    assert float((got.cpu().double() - torch.nn.functional.silu(ref)).abs().max()) <= 1e-2 * scale
    # operands that ARE bf16 numbers come through exactly up to fp32 accumulation order
    Ab = torch.from_numpy(A).bfloat16().float()
    Bb = torch.from_numpy(B).bfloat16().float()
    got2 = _C.gemm_bf16(layout, Ab.cuda(), Bb.cuda())
    ref2 = (Ab.double().T if layout == _C.GEMM_TN else Ab.double()) @ (Bb.double().T if layout == _C.GEMM_NT else Bb.double())
    assert float((got2.cpu().double() - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())
    # accumulate and the dropout-mask epilogue
    mask = (torch.from_numpy(fill.uniform((M, N), 14, 0, 1)) > 0.3).float()
    out = torch.ones(M, N, device="cuda")
    _C.gemm_bf16(layout, Ab.cuda(), Bb.cuda(), out=out, mask=mask.cuda(), mask_scale=1.5, accumulate=True)
    assert float((out.cpu().double() - (1.0 + ref2 * mask.double() * 1.5)).abs().max()) <= 2e-5 * float(ref2.abs().max()) * 1.5 + 1e-6


def test_bf16_step_deviation_from_the_fp32_step_is_bounded():
    """B = 8192, 3x256, untagged: the same step in both precisions.  Reported (pytest -s) and bounded: ids agree on >= 97 % of the
    items' full tuples, the loss deviates by < 1e-3 relative, every gradient's norm by < 5 %."""
    import hidvae_amd  # noqa: F401
    from hidvae_amd import _C
    cfg = O.Cfg(commitment_weight=0.4, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0)
    P = O.formula_params(cfg, seed=100, with_tags=True)
    x, _, _ = O.formula_batch(cfg, 8192, seed=7, tagged=False)
    batch = types.SimpleNamespace(x=x.cuda())

    def run(prec):
        _C.set_gemm_precision(prec)
        try:
            m = build_model(cfg, P).train()
            out = m(batch, gumbel_t=0.2)
            out.loss.backward()
            with torch.no_grad():
                ids = m.get_semantic_ids(m.encode(batch.x), None, None, 0.2).sem_ids
            return float(out.loss.detach()), ids.cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            _C.set_gemm_precision("f32")

    l16, ids16, g16 = run("bf16")
    l32, ids32, g32 = run("f32")
    agree_items = float((ids16 == ids32).all(dim=1).float().mean())
    agree_level0 = float((ids16[:, 0] == ids32[:, 0]).float().mean())
    dl = abs(l16 - l32) / abs(l32)
    worst = max(abs(float(g16[k].norm()) - float(g32[k].norm())) / max(float(g32[k].norm()), 1e-12) for k in g32)
    print(f"[bf16 mode] B=8192: id tuples equal on {agree_items:.4f} of the items (level 0: {agree_level0:.4f}), loss {l16:.6f} vs {l32:.6f} "
          f"(rel {dl:.2e}), worst gradient-norm deviation {worst:.3f}")
    assert agree_items >= 0.97 and dl < 1e-3 and worst < 0.05
