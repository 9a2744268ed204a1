"""exact-chain forward layers: gemm_tile16_kernel (HIDVAE_GEMM_T16=1: 32x64 tiles, =2: 64x64) against gemm_directL16_kernel (=0): bits and time"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
tag = f"T16={os.environ.get('HIDVAE_GEMM_T16', '1')}"
torch.manual_seed(0)
import zlib
for M, N, K in [(1024, 512, 768), (1024, 768, 512), (1000, 300, 333), (2048, 512, 768), (2048, 768, 512), (4096, 512, 768), (1024, 256, 512), (1024, 512, 256),
                (1024, 520, 200), (2048, 256, 512)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda"); aux = torch.empty(M, N, device="cuda")
    _C.gemm(_C.GEMM_NT, A, W, bias=b, out=out, epilogue=_C.EPI_SILU, aux=aux, split_k=1)
    crc = zlib.crc32(out.cpu().numpy().tobytes()) ^ zlib.crc32(aux.cpu().numpy().tobytes())
    err = float((aux.double() - (A.double() @ W.double().T + b.double())).abs().max())
    t = bench.time_kernel(lambda: _C.gemm(_C.GEMM_NT, A, W, bias=b, out=out, epilogue=_C.EPI_SILU, aux=aux, split_k=1))
    print(f"{tag} NT {M}x{N}x{K}: {t:6.1f} us  {2.0 * M * N * K / t * 1e-6:6.1f} TFLOP/s  crc {crc:08x}  max err {err:.2e}", flush=True)
