import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = "L16" if os.environ.get("HIDVAE_GEMM_L16", "0") == "1" else "cur"
for M, N, K in [(1024, 512, 768), (1024, 256, 512), (1024, 768, 512), (1024, 512, 256), (1024, 128, 256), (2048, 512, 768), (1000, 300, 333)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda"); pre = torch.empty(M, N, device="cuda")
    t = bench.time_kernel(lambda: _C.gemm(_C.GEMM_NT, A, B, out=out, epilogue=_C.EPI_SILU, aux=pre, split_k=1))
    print(f"{tag} NT {M}x{N}x{K} split_k=1: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s", flush=True)
