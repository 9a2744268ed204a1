"""time the forward / NN GEMMs of the step's shapes (run once with HIDVAE_GEMM_LDS=0 and once with =1)"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = "LDS" if os.environ.get("HIDVAE_GEMM_LDS", "0") == "1" else "reg"
shapes = [("NT", 1024, 512, 768, 1), ("NT", 1024, 256, 512, 1), ("NT", 1024, 768, 512, 1), ("NT", 1024, 512, 256, 1),
          ("NT", 1024, 691, 768, 0), ("NT", 1024, 768, 691, 0), ("NT", 1024, 460, 512, 0), ("NT", 1024, 512, 768, 0),
          ("NN", 1024, 512, 768, 0), ("NN", 1024, 768, 691, 0), ("NT", 2048, 768, 512, 1), ("NT", 4096, 256, 512, 1)]
for lay, M, N, K, sk in shapes:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if lay == "NT" else torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    L = _C.GEMM_NT if lay == "NT" else _C.GEMM_NN
    t = bench.time_kernel(lambda: _C.gemm(L, A, B, out=out, split_k=sk))
    print(f"{tag} {lay} {M}x{N}x{K} split_k={sk}: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s", flush=True)
