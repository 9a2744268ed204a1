import sys, types
sys.path.insert(0, ".")
import torch
from oracle import torch_oracle as O
from tests.test_model_gpu import build_model
from tests.test_dp_gpu import CFG, _batch
cfg = O.Cfg(**CFG)
m = build_model(cfg, O.formula_params(cfg, seed=100, with_tags=True)).train()
m.dp_cut = True
out = m(_batch(cfg, 128, 500, False), gumbel_t=0.2)
m.dp_cut = False
seen = set()
def walk(fn, depth=0):
    if fn is None or id(fn) in seen: return
    seen.add(id(fn))
    print("  " * depth + type(fn).__name__)
    for nf, _ in fn.next_functions:
        walk(nf, depth + 1)
walk(out.loss.grad_fn)
print("cut pairs:", [(type(t.grad_fn).__name__, tuple(t.shape)) for t, _ in m._cut_pairs])
out.loss.backward()
print("piece 1 ok; leaf grads:", [l.grad is not None for _, l in m._cut_pairs])
m.backward_rest()
print("piece 2 ok")
