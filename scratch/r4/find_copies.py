# which torch-level copies / clones run inside one eager train step?
import sys, types, traceback, collections
sys.path.insert(0, "/root/repo")
import torch, hidvae_amd
import bench
tg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sys.argv = sys.argv[:1]
args = bench.parse()
args.tagged = tg
dev = torch.device("cuda:0")
from hidvae_amd.optim import HidvaeAdamW
from hidvae_amd.step import GraphedTrainStep
m = bench.build_model(args, dev)
opt = HidvaeAdamW(bench.param_groups(m, tagged=bool(args.tagged)), cosine=(400000, 7e-8)).prepare()
px, pte, pti = bench.synth_pool(args, dev, 0)
def pb(i):
    b = types.SimpleNamespace(x=px[i % args.pool])
    if args.tagged:
        b.tags_emb, b.tags_indices = pte[i % args.pool], pti[i % args.pool]
    return b
st = GraphedTrainStep(m, opt, [pb(0)], gumbel_t=0.2, warmup=3, enabled=False)
for i in range(3):
    st([pb(i)])
torch.cuda.synchronize()
seen = collections.Counter()
from torch.utils._python_dispatch import TorchDispatchMode
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types_, args=(), kwargs=None):
        name = str(func)
        fr = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename and "find_copies" not in f.filename]
        where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno}" if fr else "?"
        seen[(name, where)] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    st([pb(5)])
torch.cuda.synchronize()
for (n, w), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    if any(k in n for k in ("copy", "clone", "contiguous", "fill", "zero", "add", "mul", "cat", "stack", "to.", "_to_copy", "index", "select", "t.default", "transpose")):
        print(c, n, w)
