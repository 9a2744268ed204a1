"""How much of the step is the gap between graph replays?  (a) the bench loop (copy + replay), (b) replay only, (c) U steps per graph."""
import os, sys, time, types
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench
sys.argv = [sys.argv[0]]
args = bench.parse()
device = torch.device("cuda:0")
from hidvae_amd.optim import HidvaeAdamW
from hidvae_amd.step import GraphedTrainStep
m = bench.build_model(args, device)
opt = HidvaeAdamW(bench.param_groups(m, tagged=False), cosine=(400000, 7e-8)).prepare()
pool_x, _, _ = bench.synth_pool(args, device, 0)
pb = lambda i: types.SimpleNamespace(x=pool_x[i % args.pool])
st = GraphedTrainStep(m, opt, [pb(0)], gumbel_t=0.2, warmup=3)
for i in range(6): st([pb(i)])
def timeit(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
k = [0]
def a():
    k[0] += 1; st([pb(k[0])])
print(f"(a) copy + replay       : {timeit(a):7.1f} us/step")
g1 = st.graphs[0]
print(f"(b) replay only         : {timeit(g1.replay):7.1f} us/step")
for U in (2, 4, 8):
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, pool=g1.pool()):
        for _ in range(U): st._eager()
    print(f"(c) {U} steps per graph   : {timeit(g.replay, 100) / U:7.1f} us/step")
    gc = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc, pool=g1.pool()):
        for u in range(U):
            st.static[0].x.copy_(pool_x[u % args.pool]); st._eager()
    print(f"(d) {U} x (copy + step)   : {timeit(gc.replay, 100) / U:7.1f} us/step")
