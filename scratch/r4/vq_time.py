"""time the corpus VQ launches (1,048,576 items, 3 x 256): training form and ids-only"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
N, L, K = 1 << 20, 3, 256
y = torch.randn(N, 32, device="cuda")
tabs = [((torch.rand(K, 32, device="cuda") * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i)) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [True, False, False])
t = bench.time_kernel(lambda: _C.rq_forward(y, cb, cc, True, _C.MODE_ROTATION, True, 0.4), launches=4, reps=10)
print(f"training form: {t:.1f} us  ({540.0 * N / t * 1e-3 / 8000:.3f} of 8 TB/s on 540 B/item)")
t = bench.time_kernel(lambda: _C.rq_ids(y, cb, cc, True), launches=4, reps=10)
print(f"ids only:      {t:.1f} us  ({152.0 * N / t * 1e-3 / 8000:.3f} of 8 TB/s on 152 B/item)")
