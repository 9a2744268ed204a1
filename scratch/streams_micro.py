import sys, time
sys.path.insert(0, '.')
import torch, hidvae_amd
from hidvae_amd import _C as C
dev = torch.device('cuda')
x = [torch.randn(1024, 512, device=dev) for _ in range(3)]
w = [torch.randn(512, 512, device=dev) * 0.05 for _ in range(3)]
g = [torch.ones(512, device=dev) for _ in range(3)]
def chain(i, n):
    h = x[i]
    for _ in range(n):
        h = C.layernorm_fwd(h, g[i], g[i], 1e-5, False, None, 1.0, None)[0]
    return h
def run(streams, n=60):
    if streams:
        sts = [torch.cuda.Stream() for _ in range(2)]
    def body():
        main = torch.cuda.current_stream()
        if not streams:
            for i in range(3): chain(i, n)
        else:
            for st in sts: st.wait_stream(main)
            chain(0, n)
            for i, st in enumerate(sts):
                with torch.cuda.stream(st): chain(i + 1, n)
            for st in sts: main.wait_stream(st)
    body(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): body()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
print("1 stream, 180 small kernels: %.1f us" % run(False))
print("3 streams x 60 kernels:      %.1f us" % run(True))
