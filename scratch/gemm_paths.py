"""fwd-shape GEMM timing + exactness check vs the C oracle for the path selected by HIDVAE_GEMM_PATH."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import hidvae_amd
from hidvae_amd import _C
from oracle import exact
dev = torch.device('cuda')

def bench(fn, n=20, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / n * 1e6

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
print("path", os.environ.get("HIDVAE_GEMM_PATH", "0"), "B", B)
shapes = [(768, 512), (512, 256), (256, 128), (128, 32), (32, 128), (128, 256), (256, 512), (512, 768)]
for (k, n) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; out = torch.empty(B, n, device=dev); aux = torch.empty(B, n, device=dev)
    us = bench(lambda: _C.gemm(_C.GEMM_NT, x, w, out=out, epilogue=_C.EPI_SILU, aux=aux))
    ref = exact.linear(x.cpu().numpy(), w.cpu().numpy(), True)
    ok = None
    if ref is not None:
        ok = np.array_equal(out.cpu().numpy(), ref)
    print(f"NT M={B} N={n} K={k}: {us:7.2f} us  {2*B*n*k/us*1e-6:6.1f} TF  exact={ok}")
    if mode == 'all':
        g = torch.randn(B, n, device=dev); gx = torch.empty(B, k, device=dev); gw = torch.empty(n, k, device=dev)
        u1 = bench(lambda: _C.gemm(_C.GEMM_NN, g, w, out=gx, split_k=0))
        r1 = (g.double() @ w.double()).float()
        u2 = bench(lambda: _C.gemm(_C.GEMM_TN, g, x, out=gw, split_k=0))
        r2 = (g.double().T @ x.double()).float()
        print(f"   NN dX {u1:7.2f} us err {float((gx-r1).abs().max()/r1.abs().max()):.1e} | TN dW {u2:7.2f} us err {float((gw-r2).abs().max()/r2.abs().max()):.1e}")
