"""Which part of the corpus-sized VQ launch costs what: the same launch with output pointers left out (C ABI called directly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hidvae_amd
from hidvae_amd import _C
dev = torch.device('cuda')
torch.manual_seed(0)
L, K, B = 3, 256, 1 << 20
tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [i == 0 for i in range(L)])
y = torch.randn(B, 32, device=dev)
z = torch.empty(B, 32, device=dev); ids = torch.empty(B, L, device=dev, dtype=torch.int64)
cat = torch.empty(B, L * 32, device=dev); esum = torch.empty(B, 32, device=dev); ql = torch.empty(B, device=dev)
lib, p = _C.lib(), _C._p
variants = {"full": (z, cat, esum, ql), "no_emb_cat": (z, None, esum, ql), "ids_only": (None, None, None, None),
            "cat_only": (None, cat, None, None)}
only = sys.argv[1:] or list(variants)
for train in (1, 0):
    for name in only:
        zz, cc_, es, q = variants[name]
        def run():
            rc = lib.hidvae_rq_forward(p(y), B, 1, p(cb), p(cc), L, K, 3, train, 0.4, p(zz), p(ids), p(cc_), L * 32, p(es), None, p(q), _C._stream())
            assert rc == 0, lib.hidvae_last_error()
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"train={train} {name:12s} {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
