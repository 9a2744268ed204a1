# absolute start / end of every launch of the stamped tagged step (LaunchStamps.spans), one row per launch: where lane 0 waits
import sys, types
sys.path.insert(0, "/root/repo")
sys.argv = sys.argv[:1]
import torch, hidvae_amd
import bench
from hidvae_amd import _C
args = bench.parse(); args.tagged = 1
dev = torch.device("cuda:0")
from hidvae_amd.optim import HidvaeAdamW
from hidvae_amd.step import GraphedTrainStep
m = bench.build_model(args, dev)
opt = HidvaeAdamW(bench.param_groups(m, tagged=True), cosine=(400000, 7e-8)).prepare()
px, pte, pti = bench.synth_pool(args, dev, 0)
def pb(i):
    return types.SimpleNamespace(x=px[i % args.pool], tags_emb=pte[i % args.pool], tags_indices=pti[i % args.pool])
st = GraphedTrainStep(m, opt, [pb(0)], gumbel_t=0.2, warmup=3)
for i in range(6):
    st([pb(i)])
torch.cuda.synchronize()
inner = st._fwd_bwd
st.graphs = None
stamps = _C.stamps_begin(dev)
try:
    st([pb(0)])
finally:
    _C.stamps_end()
for i in range(5):
    st([pb(1 + i)])
torch.cuda.synchronize()
sp = stamps.spans()
lanes = {}
for n, s, a, b in sp:
    lanes.setdefault(s, len(lanes))
prev_end = {}
for i, (n, s, a, b) in enumerate(sp):
    l = lanes[s]
    gap = a - prev_end.get(l, a)
    prev_end[l] = b
    print(f"{i:4d} lane {l} {n:36s} start {a:8.2f} end {b:8.2f} dur {b - a:6.2f} gap-on-lane {gap:6.2f}")
