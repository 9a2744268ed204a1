"""How much does one tiny kernel cost inside a replayed HIP graph (un-profiled)?"""
import sys, time
sys.path.insert(0, '.')
import torch
import hidvae_amd
from hidvae_amd import _C
dev = torch.device('cuda')
x = torch.randn(64, 32, device=dev)
w = torch.randn(32, 32, device=dev)
out = torch.empty(64, 32, device=dev)
def tiny(n):
    for _ in range(n):
        _C.gemm(_C.GEMM_NT, x, w, out=out)
for n in (10, 100):
    tiny(3); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tiny(n)
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"graph of {n} tiny gemms: {dt*1e6:.1f} us per replay -> {dt*1e6/n:.2f} us per kernel")
# eager back-to-back
tiny(10); torch.cuda.synchronize()
t0 = time.perf_counter(); tiny(200); torch.cuda.synchronize()
print(f"eager: {(time.perf_counter()-t0)/200*1e6:.2f} us per kernel (host-bound?)")
