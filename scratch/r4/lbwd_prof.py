"""eager launches of hidvae_linear_bwd on one shape, for rocprofv3 passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hidvae_amd
from hidvae_amd import _C
B, n_out, n_in = [int(v) for v in os.environ.get("SHAPE", "1024,691,768").split(",")]
g = torch.randn(B, n_out, device="cuda"); x = torch.randn(B, n_in, device="cuda"); w = torch.randn(n_out, n_in, device="cuda")
y = torch.relu(torch.randn(B, n_in, device="cuda"))
for _ in range(12):
    _C.linear_bwd(g, x, w, True, _C.EPI_DRELU, y, bias=True, dx_scale=1.25)
torch.cuda.synchronize()
