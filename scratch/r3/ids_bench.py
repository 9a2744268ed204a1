"""ids-only corpus search: time per launch at 1M / 256k / 64k items (HIDVAE_RQ_IDS_NW picks the waves per workgroup)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hidvae_amd import _C
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L, K = 3, 256
g = torch.Generator(device="cuda").manual_seed(0)
tabs = [(torch.rand(K, 32, device=dev, generator=g) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [True, False, False])
for n in (1 << 20, 1 << 18, 1 << 16):
    y = torch.randn(n, 32, device=dev, generator=g)
    t_ids = bench.time_kernel(lambda: _C.rq_ids(y, cb, cc, True), launches=2, reps=5)
    t_eval = bench.time_kernel(lambda: _C.rq_forward(y, cb, cc, True, 2, False, 0.4), launches=2, reps=5)
    t_train = bench.time_kernel(lambda: _C.rq_forward(y, cb, cc, True, 3, True, 0.4), launches=2, reps=5)
    print(f"NW={os.environ.get('HIDVAE_RQ_IDS_NW', 'default')} items={n}: ids-only {t_ids:.1f} us ({(152 * n) / t_ids * 1e-6:.3f} TB/s of 152 B/item), "
          f"eval full {t_eval:.1f} us, training {t_train:.1f} us", flush=True)
