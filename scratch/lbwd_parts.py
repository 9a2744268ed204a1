"""the parts of one Linear backward timed alone: dW (TN), dX (NN), db (column sums), and the paired launch"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
for B, no, ni in [(1024, 768, 512), (1024, 512, 256), (1024, 691, 768), (1024, 460, 512), (1024, 256, 128)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda"); dX = torch.empty(B, ni, device="cuda"); db = torch.empty(no, device="cuda")
    t_tn = bench.time_kernel(lambda: _C.gemm(_C.GEMM_TN, g, x, out=dW, split_k=0))
    t_nn = bench.time_kernel(lambda: _C.gemm(_C.GEMM_NN, g, w, out=dX, split_k=0))
    t_cs = bench.time_kernel(lambda: _C.colsum(g, out=db))
    t_p = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=True, dW=dW, bias=True, db=db))
    f = 2.0 * B * no * ni
    print(f"B={B} {no}x{ni}: TN {t_tn:5.1f} us ({f / t_tn * 1e-6:5.1f} TF)  NN {t_nn:5.1f} us ({f / t_nn * 1e-6:5.1f} TF)  colsum {t_cs:4.1f} us  "
          f"pair {t_p:5.1f} us ({2 * f / t_p * 1e-6:5.1f} TF)", flush=True)
