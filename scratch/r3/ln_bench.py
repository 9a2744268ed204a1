"""LayerNorm forward / backward-partial stand-alone (warm) at the tag heads' shapes"""
import sys
sys.path.insert(0, ".")
import torch
import hidvae_amd  # noqa: F401
from hidvae_amd import _C
import bench
for M, N in [(1024, 768), (1024, 691), (1024, 512), (1024, 256), (1024, 230), (2048, 768), (1024, 64)]:
    x = torch.randn(M, N, device="cuda"); g = torch.randn(M, N, device="cuda")
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    y, mu, rs = _C.layernorm_fwd(x, gam, bet, 1e-5, True, None, 1.0, None)
    tf = bench.time_kernel(lambda: _C.layernorm_fwd(x, gam, bet, 1e-5, True, None, 1.0, None))
    tb = bench.time_kernel(lambda: _C.layernorm_bwd_partial(g, x, gam, bet, mu, rs, True, y, 1.0, 0.0))
    mb = M * N * 4e-6
    print(f"M={M} N={N}: fwd {tf:5.1f} us ({2 * mb / tf * 1e-3:4.2f} TB/s)   bwd_partial {tb:5.1f} us ({4 * mb / tb * 1e-3:4.2f} TB/s)", flush=True)
