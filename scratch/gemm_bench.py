"""Per-launch time of single kernels, replayed from a graph of 20 launches (L2-warm, same operands)."""
import sys, time
sys.path.insert(0, '.')
import torch
import hidvae_amd
from hidvae_amd import _C
dev = torch.device('cuda')

def bench(name, fn, n=20, reps=30):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps / n * 1e6
    print(f"{name:60s} {us:8.2f} us/launch")
    return us

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = [(768, 512), (512, 256), (256, 128), (128, 32), (32, 128), (128, 256), (256, 512), (512, 768)]
for (k, n) in shapes:
    x = torch.randn(B, k, device=dev); w = torch.randn(n, k, device=dev); out = torch.empty(B, n, device=dev); aux = torch.empty(B, n, device=dev)
    us = bench(f"NT fwd  M={B} N={n} K={k} (exact, silu)", lambda: _C.gemm(_C.GEMM_NT, x, w, out=out, epilogue=_C.EPI_SILU, aux=aux))
    print(f"      -> {2*B*n*k/us*1e-6:.1f} TFLOP/s, chain bound {k/16*8*64/2.39e3:.1f} us")
    g = torch.randn(B, n, device=dev); gx = torch.empty(B, k, device=dev); gw = torch.empty(n, k, device=dev)
    bench(f"NN dX   M={B} N={k} K={n} (auto split)", lambda: _C.gemm(_C.GEMM_NN, g, w, out=gx, split_k=0))
    bench(f"TN dW   M={n} N={k} K={B} (auto split)", lambda: _C.gemm(_C.GEMM_TN, g, x, out=gw, split_k=0))
y = torch.randn(B, 32, device=dev)
tabs = [torch.rand(256, 32, device=dev) for _ in range(3)]
cb, cc = _C.codebook_prepare(tabs, [True, False, False])
bench("rq_forward B=%d 3x256" % B, lambda: _C.rq_forward(y, cb, cc, True, 3, True, 0.4))
