# stand-alone duration of the batch gather (hidvae_gather_rows) against three torch.index_select launches and three device copies
import sys
sys.path.insert(0, "/root/repo")
import torch, hidvae_amd
import bench
from hidvae_amd import _C
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B in (1024, 2048, 8192):
    n = 8 * B
    x, te, ti = torch.randn(n, 768).to(dev), torch.randn(n, 3, 768).to(dev), torch.randint(0, 40, (n, 3)).to(dev)
    idx = torch.randperm(n, generator=g)[:B].to(dev)
    outs = [torch.empty(B, 768, device=dev), torch.empty(B, 3, 768, device=dev), torch.empty(B, 3, dtype=torch.int64, device=dev)]
    t1 = bench.time_kernel(lambda: _C.gather_rows(idx, [x, te, ti], outs))
    def sel():
        torch.index_select(x, 0, idx, out=outs[0]); torch.index_select(te, 0, idx, out=outs[1]); torch.index_select(ti, 0, idx, out=outs[2])
    t2 = bench.time_kernel(sel)
    def cp():
        outs[0].copy_(x[:B]); outs[1].copy_(te[:B]); outs[2].copy_(ti[:B])
    t3 = bench.time_kernel(cp)
    mb = B * (768 * 4 * 4 + 24) / 1e6
    print(f"B={B}: gather_rows {t1:.2f} us ({2 * mb / t1:.2f} TB/s read+write), 3 x index_select {t2:.2f} us, 3 x copy_ {t3:.2f} us; {mb:.1f} MB")
