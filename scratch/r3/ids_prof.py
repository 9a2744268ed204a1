import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hidvae_amd import _C
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L, K = 3, 256
g = torch.Generator(device="cuda").manual_seed(0)
tabs = [(torch.rand(K, 32, device=dev, generator=g) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [True, False, False])
y = torch.randn(1 << 20, 32, device=dev, generator=g)
for _ in range(4):
    _C.rq_ids(y, cb, cc, True)
torch.cuda.synchronize()
