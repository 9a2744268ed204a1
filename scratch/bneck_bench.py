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
for B in (1024, 2048, 4096):
    K2, N2, Nd0, Nd1, L, K = 256, 128, 128, 256, 3, 256
    h1 = torch.randn(B, K2, device=dev)
    W2, W3 = torch.randn(N2, K2, device=dev) * 0.05, torch.randn(32, N2, device=dev) * 0.1
    Wd0, Wd1 = torch.randn(Nd0, 32, device=dev) * 0.2, torch.randn(Nd1, Nd0, device=dev) * 0.1
    tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
    cb, cc = C.codebook_prepare(tabs, [i == 0 for i in range(L)])
    t_f = bench(lambda: C.bottleneck_fwd(h1, W2, W3, cb, cc, True, 3, 0.4, Wd0, Wd1))
    def sep():
        pre2 = torch.empty(B, N2, device=dev)
        h2 = C.gemm(C.GEMM_NT, h1, W2, epilogue=C.EPI_SILU, aux=pre2)
        y = C.gemm(C.GEMM_NT, h2, W3)
        z, ids, emb_cat, emb_sum, _, qloss = C.rq_forward(y, cb, cc, True, 3, True, 0.4)
        pre_d0 = torch.empty(B, Nd0, device=dev)
        d0 = C.gemm(C.GEMM_NT, emb_sum, Wd0, epilogue=C.EPI_SILU, aux=pre_d0)
        pre_d1 = torch.empty(B, Nd1, device=dev)
        d1 = C.gemm(C.GEMM_NT, d0, Wd1, epilogue=C.EPI_SILU, aux=pre_d1)
    t_s = bench(sep)
    print(f"B={B}: fused {t_f:.2f} us, separate {t_s:.2f} us")
