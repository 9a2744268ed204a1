import sys, types
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import torch_oracle as O
from tests.test_model_gpu import build_model, make_batch
from hidvae_amd.optim import HidvaeAdamW
cfg = O.Cfg(commitment_weight=0.4, sem_id_uniqueness_weight=1.5, sem_id_uniqueness_margin=0.0)
P = O.formula_params(cfg, seed=100, with_tags=True)
m = build_model(cfg, P).train()
core = [p for n, p in m.named_parameters() if not n.startswith("tag_")]
names = [n for n, p in m.named_parameters() if not n.startswith("tag_")]
opt = HidvaeAdamW([{"params": core, "lr": 2.8e-4, "weight_decay": 0.015}], cosine=(1000, 7e-8))
Pc = {k: v.clone() for k, v in P.items() if not k.startswith("tag_")}
M = {k: torch.zeros_like(v) for k, v in Pc.items()}
V = {k: torch.zeros_like(v) for k, v in Pc.items()}
for it in range(2):
    x, _, _ = O.formula_batch(cfg, 96, seed=500 + it, tagged=False)
    opt.zero_grad()
    out = m(make_batch(x, None, None), gumbel_t=0.2)
    out.loss.backward()
    o, g = O.grads(Pc, cfg, x, training=True)
    for n, p in zip(names, core):
        a = p.grad.cpu().double(); b = g[n].double()
        print(it, n, 'grad relerr', float((a-b).abs().max()/b.abs().max()), 'gmax', float(b.abs().max()), 'gmin', float(b.abs().min()))
    opt.step()
    lr = O.cosine_lr(2.8e-4, 7e-8, it, 1000)
    for k in Pc:
        Pc[k], M[k], V[k] = O.adamw_step(Pc[k], g[k], M[k], V[k], it + 1, lr, 0.015)
    sd = m.state_dict()
    for k in Pc:
        d = (sd[k].cpu()-Pc[k]).abs()
        i = d.argmax()
        print(it, k, 'param maxabs diff', float(d.max()), 'lr', lr, 'g at argmax', float(g[k].reshape(-1)[i]))
