"""experiment: Linear backward at B=1024 -- one paired direct launch vs LDS-tiled kernels with grid-level split-K"""
import os, sys, ctypes
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
B, n_out, n_in = 1024, 768, 512
g = torch.randn(B, n_out, device="cuda"); x = torch.randn(B, n_in, device="cuda"); w = torch.randn(n_out, n_in, device="cuda") * 0.03
pre = torch.randn(B, n_in, device="cuda")
t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, True, _C.EPI_DSILU, pre))
print(f"paired direct launch: {t:.1f} us")
def tiled(sp_w, sp_x):
    dW = torch.empty(n_out, n_in, device="cuda"); dX = torch.empty(B, n_in, device="cuda")
    wsw = torch.empty(sp_w * n_out * n_in, device="cuda"); wsx = torch.empty(max(1, sp_x) * B * n_in, device="cuda")
    L = _C.lib()
    def run():
        _C._check(L.hidvae_gemm_f32(_C.GEMM_TN, n_out, n_in, B, _C._p(g), n_out, _C._p(x), n_in, None, _C._p(dW), n_in, 0, None, 0, None, 0, 1.0, sp_w,
                                    _C._p(wsw), 0, _C._stream()), "tn")
        _C._check(L.hidvae_gemm_f32(_C.GEMM_NN, B, n_in, n_out, _C._p(g), n_out, _C._p(w), n_in, None, _C._p(dX), n_in, _C.EPI_DSILU, _C._p(pre), n_in, None, 0,
                                    1.0, sp_x, _C._p(wsx), 0, _C._stream()), "nn")
    return bench.time_kernel(run)
for sp_w, sp_x in ((4, 2), (8, 4), (8, 2), (16, 4), (4, 4)):
    print(f"tiled split dW {sp_w} dX {sp_x}: {tiled(sp_w, sp_x):.1f} us (both GEMMs + their reduces)")
