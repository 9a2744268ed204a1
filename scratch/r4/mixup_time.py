# stand-alone duration of hidvae_mixup_plan at the step's sizes
import sys
sys.path.insert(0, "/root/repo")
import torch, hidvae_amd
import bench
from hidvae_amd import _C
from hidvae_amd.rand import DeviceRand
dev = torch.device("cuda:0")
for B in (1024, 2048, 8192):
    t = torch.randint(0, 38, (B, 3), device=dev)
    r = DeviceRand(0.2, seed=1)
    st = r.state(dev)
    us = bench.time_kernel(lambda: _C.mixup_plan(t, None, 0.2, rng_state=st))
    print(f"B={B}: mixup_plan {us:.2f} us")
