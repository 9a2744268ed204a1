import os, sys
sys.path.insert(0, ".")
import torch, hidvae_amd
from hidvae_amd import _C
import bench
for B, no, ni in [(8192, 32, 128), (8192, 128, 32), (8192, 128, 256), (8192, 256, 128), (4096, 32, 128), (4096, 128, 32)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda"); dX = torch.empty(B, ni, device="cuda"); db = torch.empty(no, device="cuda")
    t_tn = bench.time_kernel(lambda: _C.gemm(_C.GEMM_TN, g, x, out=dW, split_k=0))
    t_nn = bench.time_kernel(lambda: _C.gemm(_C.GEMM_NN, g, w, out=dX, split_k=0))
    t_cs = bench.time_kernel(lambda: _C.colsum(g, out=db))
    t_p = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=True, dW=dW, bias=False))
    print(f"B={B} {no}x{ni}: TN {t_tn:5.1f} us  NN {t_nn:5.1f} us  colsum {t_cs:4.1f} us  linear_bwd(no bias) {t_p:5.1f} us", flush=True)
