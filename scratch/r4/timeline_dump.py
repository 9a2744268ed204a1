# every launch of the tagged step in issue order with its in-step duration (bench.step_timeline)
import sys, types
sys.path.insert(0, "/root/repo")
tg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.argv = sys.argv[:1]
import torch, hidvae_amd
import bench
args = bench.parse(); args.tagged = tg
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
st = GraphedTrainStep(m, opt, [pb(0)], gumbel_t=0.2, warmup=3)
for i in range(6):
    st([pb(i)])
torch.cuda.synchronize()
rows, empty = bench.step_timeline(st, pb, dev)
print(f"# empty bracket {empty:.2f} us; {len(rows)} launches; sum {sum(r['us'] for r in rows):.1f} us")
for i, r in enumerate(rows):
    d = f"{r['M']}x{r['N']}x{r['K']}" if "M" in r else ""
    print(f"{i:4d} {r['entry']:38s} {r['us']:7.2f} {d}")
