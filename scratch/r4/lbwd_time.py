"""time hidvae_linear_bwd on the step's shapes (graph of 20 launches, HIP events); HIDVAE_RING selects the kernel"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hidvae_amd
from hidvae_amd import _C
import bench

shapes = [(1024, 691, 768)] if os.environ.get('ONE') else [(1024, 768, 691), (1024, 691, 768), (1024, 768, 512), (1024, 512, 768), (1024, 460, 512), (1024, 512, 460), (1024, 345, 691),
          (1024, 512, 256), (1024, 256, 230), (1024, 230, 256), (2048, 768, 691), (2048, 512, 460), (4096, 768, 512), (8192, 512, 256)]
side = torch.cuda.Stream()
_C.register_ws_lane(side)
out = {}
for B, n_out, n_in in shapes:
    g = torch.randn(B, n_out, device="cuda"); x = torch.randn(B, n_in, device="cuda"); w = torch.randn(n_out, n_in, device="cuda")
    y = torch.relu(torch.randn(B, n_in, device="cuda"))
    fn = lambda: _C.linear_bwd(g, x, w, True, _C.EPI_DRELU, y, bias=True, dx_scale=1.25)
    fl = 4.0 * B * n_out * n_in
    us_main = bench.time_kernel(fn)
    us_side = bench.time_kernel(fn, side=side)
    out[f"{B}x{n_out}x{n_in}"] = dict(us_main=round(us_main, 2), frac_main=round(fl / us_main / 1e6 / 157.3, 3), us_side=round(us_side, 2),
                                       frac_side=round(fl / us_side / 1e6 / 157.3, 3))
    print(f"{B:5d} x {n_out:4d} x {n_in:4d}: main {us_main:7.2f} us ({fl / us_main / 1e6 / 157.3:.3f})   side {us_side:7.2f} us ({fl / us_side / 1e6 / 157.3:.3f})", flush=True)
print(json.dumps(out))
