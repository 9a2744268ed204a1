import sys; sys.path.insert(0,'.')
import torch
from oracle import torch_oracle as O
from tests.test_model_gpu import build_model, make_batch
from hidvae_amd.rand import InjectedRand
import hidvae_amd.ops as ops
cfg = O.Cfg(tag_class_counts=[38, 168, 348], use_focal_loss=True, focal_loss_params={"gamma_0": 2.7, "alpha_0": 0.24}, commitment_weight=0.4)
P = O.formula_params(cfg, seed=100, with_tags=True)
B = int(sys.argv[1]) if len(sys.argv)>1 else 2048
x, te, ti = O.formula_batch(cfg, B, seed=3, tagged=True)
def run():
    m = build_model(cfg, P).train(); m.rand = InjectedRand(O.FormulaRand())
    out = m(make_batch(x, te, ti), gumbel_t=0.2); out.loss.backward(); torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
for trial in range(3):
    a, b = run(), run()
    bad = [(k, float((a[k]-b[k]).abs().max()), float(a[k].abs().max())) for k in a if not torch.equal(a[k], b[k])]
    print('trial', trial, 'differing keys:', len(bad))
    for k, d, m in bad[:12]: print('   ', k, d, m)
if len(sys.argv) > 2:
    # serialize: no side stream
    ops._SIDE.clear()
    import torch.cuda
    ops.side_stream = lambda: torch.cuda.current_stream()
