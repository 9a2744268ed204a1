import sys, time
sys.path.insert(0, '.')
import torch
import hidvae_amd
from hidvae_amd import _C
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
B = 1024
for (k, n) in [(768, 512), (512, 256), (256, 128), (128, 32), (32, 128), (128, 256), (256, 512), (512, 768)]:
    x = torch.randn(B, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; out = torch.empty(B, n, device=dev); aux = torch.empty(B, n, device=dev)
    r = []
    for sk in (1, 2, 4, 0):
        r.append(bench(lambda: _C.gemm(_C.GEMM_NT, x, w, out=out, epilogue=_C.EPI_SILU, aux=aux, split_k=sk)))
    print(f"NT M={B} N={n} K={k}: exact {r[0]:6.2f}  split2 {r[1]:6.2f}  split4 {r[2]:6.2f}  auto {r[3]:6.2f}")
