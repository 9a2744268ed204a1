"""How fast does this chip take plain stores?  (context for the VQ launch's 700 MB output stream)"""
import torch
dev = torch.device('cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (100, 400, 700, 2000):
    x = torch.empty(mb * 250_000, device=dev)
    y = torch.empty_like(x)
    us = timeit(lambda: x.fill_(1.0))
    print(f"fill  {mb:5d} MB: {us:8.1f} us  {mb / us * 1e-3 * 1e3:6.2f} TB/s... {mb*1e6/us*1e-6:6.2f} TB/s", flush=True)
    us = timeit(lambda: y.copy_(x))
    print(f"copy  {mb:5d} MB: {us:8.1f} us  read+write {2*mb*1e6/us*1e-6:6.2f} TB/s", flush=True)
    us = timeit(lambda: torch.sum(x))
    print(f"sum   {mb:5d} MB: {us:8.1f} us  read {mb*1e6/us*1e-6:6.2f} TB/s", flush=True)
