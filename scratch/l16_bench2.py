import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = "L16=" + os.environ.get("HIDVAE_GEMM_L16", "0")
for M, N, K in [(1024, 691, 768), (1024, 768, 691), (1024, 460, 512), (1024, 512, 460), (1024, 230, 256), (1024, 256, 230), (1024, 512, 768), (1024, 345, 691), (1024, 96, 512), (2048, 691, 768)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    t = bench.time_kernel(lambda: _C.gemm(_C.GEMM_NT, A, B, out=out, bias=bias, epilogue=_C.EPI_RELU, split_k=0))
    print(f"{tag} NT {M}x{N}x{K} split_k=0: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s", flush=True)
