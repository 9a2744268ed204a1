import sys, time
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd import _C
dev = torch.device('cuda')
torch.manual_seed(0)
L, K = 3, 256
tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [i == 0 for i in range(L)])
for B in (1 << 17, 1 << 18, 1 << 19, 1 << 20):
    y = torch.randn(B, 32, device=dev)
    for train in (True, False):
        for _ in range(3): out = _C.rq_forward(y, cb, cc, True, 3, train, 0.4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = _C.rq_forward(y, cb, cc, True, 3, train, 0.4)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"B={B} train={train}: {us:8.1f} us  {B/us:7.1f} M items/s  mfma_frac {2.0*B*L*K*32/us*1e-6/157.3:.3f}  ids_sum {int(out[1].sum())}")
