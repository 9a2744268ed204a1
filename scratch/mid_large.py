"""Linear backward at large batches: the balanced kernel (HIDVAE_GEMM_MID_MAX_B=16384) against the LDS-tiled + K-slab path"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = f"MAX_B={os.environ.get('HIDVAE_GEMM_MID_MAX_B', '4095')}"
for B, no, ni, dx in [(4096, 768, 512, 1), (4096, 512, 256, 1), (4096, 512, 768, 0), (4096, 256, 128, 1), (8192, 768, 512, 1), (8192, 512, 256, 1),
                      (8192, 512, 768, 0), (8192, 256, 128, 1), (8192, 128, 32, 1)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda"); db = torch.empty(no, device="cuda")
    t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=bool(dx), dW=dW, bias=True, db=db))
    fl = 2.0 * B * no * ni * (2 if dx else 1)
    print(f"{tag} B={B} {no}x{ni} dx={dx}: {t:6.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s", flush=True)
