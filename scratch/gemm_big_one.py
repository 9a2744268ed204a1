import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hidvae_amd
from hidvae_amd import _C
dev = torch.device('cuda')
B = 65536
x = torch.randn(B, 768, device=dev); w = torch.randn(512, 768, device=dev) * 0.05; out = torch.empty(B, 512, device=dev); aux = torch.empty(B, 512, device=dev)
for _ in range(5):
    _C.gemm(_C.GEMM_NT, x, w, out=out, epilogue=_C.EPI_SILU, aux=aux)
torch.cuda.synchronize()
print("done")
