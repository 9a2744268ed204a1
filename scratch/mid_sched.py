"""balanced kernel: tile-per-workgroup vs even ranges vs the paired kernels, B = 1024 .. 4096 (run once per HIDVAE_GEMM_MID / _SCHED setting)"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = f"MID={os.environ.get('HIDVAE_GEMM_MID', 'd')} SCHED={os.environ.get('HIDVAE_GEMM_MID_SCHED', '0')}"
for B, no, ni, dx in [(1024, 768, 512, 1), (1024, 512, 256, 1), (1024, 512, 768, 0), (1024, 691, 768, 1), (1024, 460, 512, 1), (1024, 345, 691, 1),
                      (2048, 768, 512, 1), (2048, 512, 256, 1), (2048, 512, 768, 0), (2048, 691, 768, 1), (2048, 460, 512, 1), (2048, 256, 128, 1),
                      (4095, 768, 512, 1), (4095, 512, 256, 1), (4095, 512, 768, 0), (4095, 256, 128, 1)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda"); db = torch.empty(no, device="cuda")
    t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=bool(dx), dW=dW, bias=True, db=db))
    fl = 2.0 * B * no * ni * (2 if dx else 1)
    print(f"{tag} B={B} {no}x{ni} dx={dx}: {t:6.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s", flush=True)
