import sys; sys.path.insert(0,'.')
import torch, hidvae_amd
from hidvae_amd.rand import DeviceRand
dev=torch.device('cuda')
r=DeviceRand(0.2)
t=torch.randint(-1,5,(1024,),device=dev)
def try_cap(name, fn):
    fn(); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(name,'OK')
    except Exception as e:
        print(name,'FAIL',str(e).split('\n')[0])
        torch.cuda.synchronize()
try_cap('dropout', lambda: r.dropout_keep((1024,512),0.4,dev))
try_cap('rand+where', lambda: torch.where(t>=0, torch.rand(1024,device=dev), torch.full((1024,),2.0,device=dev)))
try_cap('argsort', lambda: torch.argsort(torch.rand(1024,device=dev)))
try_cap('cumsum', lambda: torch.cumsum((t>=0).to(torch.int64),0))
try_cap('beta', lambda: r._beta.sample() if r._beta is not None else r.mixup_partner(t,dev))
try_cap('mixup_partner', lambda: r.mixup_partner(t,dev))
B=1024
valid = t>=0
keys=torch.rand(B,device=dev)
order=torch.argsort(keys)
rank=torch.cumsum(valid.to(torch.int64),0)-1
try_cap('index', lambda: order[rank.clamp(min=0)])
partner = torch.where(valid, order[rank.clamp(min=0)], torch.full_like(rank,-1))
try_cap('full+scatter', lambda: torch.full((B+1,),-1,dtype=torch.int64,device=dev).scatter_(0, torch.where(valid, partner, torch.full_like(partner,B)), torch.arange(B,device=dev)))
try_cap('arange', lambda: torch.arange(B,device=dev))
try_cap('beta.to', lambda: r._beta.sample().to(torch.float32))
try_cap('slice contiguous', lambda: torch.full((B+1,),-1,dtype=torch.int64,device=dev)[:B].contiguous())
