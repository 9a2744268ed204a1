# stand-alone time of hidvae_predictor_fwd on level 0's shapes (E = 32, hidden 256, mid 230, C = 38) at B = 1024
import sys
sys.path.insert(0, "/root/repo")
import torch, hidvae_amd
from hidvae_amd import _C
from hidvae_amd.modules.h_rqvae import TagPredictor
from hidvae_amd.rand import DeviceRand
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pred = TagPredictor(32, 38, hidden_dim=256, dropout_rate=0.4, use_batch_norm=True, layer_idx=0).cuda().train()
rand = DeviceRand(seed=5); rand.begin_step(torch.device("cuda"))
h = torch.randn(B, 32, device="cuda")
fe, blocks, cl = pred.feature_extractor, (pred.residual_block1, pred.residual_block2), pred.classifier
p = pred.dropout_p
mk = lambda n, pp=p: (rand.dropout_keep((B, n), pp, h.device), 1.0 / (1.0 - pp))
for drop in (True, False):
    d = (lambda n, pp=p: mk(n, pp)) if drop else (lambda n, pp=p: None)
    units = [dict(lin=fe[0], norm=fe[1], act2=True, drop2=d(256), carry=True)]
    for rb in blocks:
        units.append(dict(lin=rb[0], norm=rb[1], act2=True, drop2=d(230)))
        units.append(dict(lin=rb[4], norm=rb[7], act1=True, drop1=d(256), residual=True))
    units += [dict(lin=cl[0], norm=cl[1], act2=True, drop2=d(230)), dict(lin=cl[4], norm=None, act1=True, drop1=d(115, p * 0.5)), dict(lin=cl[7], norm=None)]
    t = bench.time_kernel(lambda: _C.predictor_fwd(h, units))
    print(f"predictor_fwd B={B} dropout={'on' if drop else 'off'}: {t:.1f} us", flush=True)
for n in (1, 3, 8):
    t = bench.time_kernel(lambda: _C.predictor_fwd(h, units[:n]))
    print(f"  first {n} unit(s), no dropout: {t:.1f} us", flush=True)
