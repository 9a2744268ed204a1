"""the small row kernels of a tag level stand-alone (warm): gate, BatchNorm, tag loss, InfoNCE rows -- level-2 shapes at B = 1024"""
import sys
sys.path.insert(0, ".")
import torch
import hidvae_amd  # noqa: F401
from hidvae_amd import _C
import bench
B = 1024
dev = "cuda"
for E in (32, 64, 96):
    xc = torch.randn(B, E, device=dev)
    W = [torch.randn(E // 4, E, device=dev) * 0.1, torch.randn(E // 4, device=dev), torch.randn(E // 2, E // 4, device=dev) * 0.1,
         torch.randn(E // 2, device=dev), torch.randn(E, E // 2, device=dev) * 0.1, torch.randn(E, device=dev)]
    hh, sv = _C.gate_fwd(xc, *W, True)
    tf = bench.time_kernel(lambda: _C.gate_fwd(xc, *W, True))
    tb = bench.time_kernel(lambda: _C.gate_bwd(hh, xc, W[0], W[2], W[4], True, sv))
    print(f"gate E={E}: fwd {tf:5.1f} us  bwd {tb:5.1f} us", flush=True)
x = torch.randn(B, 512, device=dev)
g = torch.randn(B, 512, device=dev)
gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
rm, rv = torch.zeros(512, device=dev), torch.ones(512, device=dev)
y, sm, sr = _C.batchnorm_fwd(x, gam, bet, 1e-5, 0.1, True, rm, rv, True, None, 1.0)
tf = bench.time_kernel(lambda: _C.batchnorm_fwd(x, gam, bet, 1e-5, 0.1, True, rm, rv, True, None, 1.0))
tb = bench.time_kernel(lambda: _C.batchnorm_bwd(g, x, gam, bet, sm, sr, True, None, 1.0, y_out=y))
print(f"batchnorm 1024x512: fwd {tf:5.1f} us  bwd {tb:5.1f} us", flush=True)
for C in (38, 168, 348):
    logits = torch.randn(B, C, device=dev)
    tgt = torch.randint(0, C, (B,), device=dev)
    t = bench.time_kernel(lambda: _C.tag_loss_fwd(logits, tgt, None, None, True, 2.7, 0.24, 0.1, 0.05, True))
    print(f"tag_loss_fwd C={C}: {t:5.1f} us", flush=True)
S = torch.randn(B, B, device=dev)
t = bench.time_kernel(lambda: _C.infonce_rows(S.clone(), 0.1, 1.0))
tc = bench.time_kernel(lambda: S.clone())
print(f"infonce_rows 1024x1024: {t - tc:5.1f} us (+ clone {tc:4.1f})", flush=True)
