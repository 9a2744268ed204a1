import os, sys
sys.path.insert(0, ".")
import torch, hidvae_amd
from hidvae_amd import _C
import bench
tag = "one" if os.environ.get("HIDVAE_LN_ONEPASS", "1") != "0" else "two"
for M, N in [(1024, 768), (1024, 691), (1024, 512), (1024, 460), (1024, 256), (1024, 230), (2048, 768), (4096, 768)]:
    x = torch.randn(M, N, device="cuda"); ga = torch.rand(N, device="cuda") + 0.5; be = torch.randn(N, device="cuda") * 0.1
    gy = torch.randn(M, N, device="cuda"); mask = (torch.rand(M, N, device="cuda") < 0.6).float()
    y, mean, rstd = _C.layernorm_fwd(x, ga, be, 1e-5, True, mask, 1 / 0.6, None)
    gg, gb = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    t = bench.time_kernel(lambda: _C.layernorm_bwd_all(gy, x, ga, be, mean, rstd, True, mask, 1 / 0.6, gg=gg, gb=gb))
    tf = bench.time_kernel(lambda: _C.layernorm_fwd(x, ga, be, 1e-5, True, mask, 1 / 0.6, None))
    print(f"{tag} LN {M}x{N}: bwd_all {t:5.1f} us   fwd {tf:5.1f} us", flush=True)
