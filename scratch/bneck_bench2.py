import sys, time
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd import _C as C
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
K2, N2, Nd0, Nd1 = 256, 128, 128, 256
h1 = torch.randn(B, K2, device=dev)
W2, W3 = torch.randn(N2, K2, device=dev) * 0.05, torch.randn(32, N2, device=dev) * 0.1
Wd0, Wd1 = torch.randn(Nd0, 32, device=dev) * 0.2, torch.randn(Nd1, Nd0, device=dev) * 0.1
for (L, K, mode) in ((1, 128, 3), (3, 256, 3), (1, 128, 2), (3, 256, 2)):
    tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) for i in range(L)]
    cb, cc = C.codebook_prepare(tabs, [i == 0 for i in range(L)])
    t_f = bench(lambda: C.bottleneck_fwd(h1, W2, W3, cb, cc, True, mode, 0.4, Wd0, Wd1))
    y = torch.randn(B, 32, device=dev)
    t_r = bench(lambda: C.rq_forward(y, cb, cc, True, mode, True, 0.4))
    print(f"L={L} K={K} mode={mode}: fused {t_f:.2f} us, rq_forward alone {t_r:.2f} us")
