"""hidvae_linear_bwd on the tag heads' widest layers: how far is the ranged schedule (tiles > 256 slots) from the whole-tile one?"""
import sys
sys.path.insert(0, ".")
import torch
import hidvae_amd  # noqa: F401
from hidvae_amd import _C
import bench
SIDE = torch.cuda.Stream()
_C.register_ws_lane(SIDE)
for B, no, ni, bias in [(1024, 691, 768, True), (1024, 691, 768, False), (1024, 704, 768, False), (1024, 640, 768, False), (1024, 512, 768, False),
                        (1024, 768, 691, True), (1024, 768, 704, False), (1024, 345, 691, True), (1024, 460, 512, True), (1024, 768, 512, False),
                        (2048, 691, 768, True), (2048, 768, 512, False)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda") * 0.03
    t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, True, bias=bias))
    ts = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, True, bias=bias), side=SIDE)
    f = 4.0 * B * no * ni
    nt = -(-no // 64) * -(-ni // 64) + -(-B // 64) * -(-ni // 64)
    print(f"B={B} {no}x{ni} bias={int(bias)}: {t:6.1f} us  {f / t * 1e-6:5.1f} TF/s   on a level stream {ts:6.1f} us  {f / ts * 1e-6:5.1f} TF/s   tiles {nt}", flush=True)
