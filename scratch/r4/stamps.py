"""one launch with in-kernel stamps (HIDVAE_RING_DBG=16), print per-step deltas for two workgroups"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hidvae_amd
from hidvae_amd import _C
B, n_out, n_in = [int(v) for v in os.environ.get("SHAPE", "1024,691,768").split(",")]
g = torch.randn(B, n_out, device="cuda"); x = torch.randn(B, n_in, device="cuda"); w = torch.randn(n_out, n_in, device="cuda")
y = torch.relu(torch.randn(B, n_in, device="cuda"))
for _ in range(5):
    _C.linear_bwd(g, x, w, True, _C.EPI_DRELU, y, bias=True, dx_scale=1.25)
torch.cuda.synchronize()
lane = _C._lane_ws(g.device, 4)
torch.cuda.synchronize()
max_slabs = 2048
st = lane[4096 + (max_slabs - 1) * 4096: 4096 + max_slabs * 4096].view(torch.int64).cpu().numpy().reshape(-1)[:2 * 64 * 8].reshape(2, 64, 8)
for wg in range(2):
    print("workgroup", wg)
    t0 = st[wg, 0, 0]
    for s in range(40):
        r = st[wg, s]
        if r[0] == 0: break
        nxt = st[wg, s + 1, 0] if st[wg, s + 1, 0] else r[4]
        print(f" step {s:2d} at {int(r[0]-t0):7d}: wait {int(r[1]-r[0]):5d} barrier {int(r[2]-r[1]):5d} issue {int(r[3]-r[2]):5d} mfma-issue {int(r[4]-r[3]):5d}  gap-to-next {int(nxt-r[4]):5d}")
