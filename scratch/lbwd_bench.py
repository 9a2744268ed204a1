"""time hidvae_linear_bwd (dW, dX, db in one paired launch) and a few forward GEMMs on the step's shapes"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = os.environ.get("TAG", "cur")
# (B, n_out, n_in, need_dx)
shapes = [(1024, 768, 512, 1), (1024, 512, 256, 1), (1024, 256, 128, 1), (1024, 128, 32, 1), (1024, 32, 128, 1), (1024, 128, 256, 1),
          (1024, 256, 512, 1), (1024, 512, 768, 0), (1024, 691, 768, 1), (1024, 768, 691, 1), (1024, 460, 512, 1), (1024, 512, 460, 1),
          (1024, 230, 256, 1), (1024, 256, 230, 1), (1024, 345, 691, 1), (2048, 768, 512, 1)]
tot = 0.0
for B, no, ni, dx in shapes:
    g = torch.randn(B, no, device="cuda")
    x = torch.randn(B, ni, device="cuda")
    w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda")
    db = torch.empty(no, device="cuda")
    t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=bool(dx), dW=dW, bias=True, db=db))
    fl = 2.0 * B * no * ni * (2 if dx else 1)
    tot += t
    print(f"{tag} lbwd B={B} {no}x{ni} dx={dx}: {t:6.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s", flush=True)
print(f"{tag} lbwd total {tot:.1f} us")
for lay, M, N, K, sk in [("NT", 1024, 691, 768, 0), ("NT", 1024, 768, 691, 0), ("NT", 1024, 460, 512, 0), ("NT", 1024, 512, 768, 0),
                         ("NN", 1024, 512, 768, 0), ("NN", 1024, 768, 691, 0), ("NT", 2048, 768, 512, 1)]:
    A = torch.randn(M, K, device="cuda")
    Bm = torch.randn(N, K, device="cuda") if lay == "NT" else torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    L = _C.GEMM_NT if lay == "NT" else _C.GEMM_NN
    t = bench.time_kernel(lambda: _C.gemm(L, A, Bm, out=out, split_k=sk))
    print(f"{tag} {lay} {M}x{N}x{K} split_k={sk}: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s", flush=True)
