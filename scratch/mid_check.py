"""gemm_mid_pair_kernel (HIDVAE_GEMM_MID=KG) against float64 torch on regular and ragged shapes, + timing"""
import os, sys
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
import bench
torch.manual_seed(0)
bad = 0
for B, no, ni in [(1024, 768, 512), (1024, 691, 768), (1000, 460, 345), (256, 64, 64), (333, 230, 257), (1024, 512, 768), (2048, 65, 130), (300, 128, 64), (1024, 256, 128), (4095, 70, 67)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    pre = torch.randn(B, ni, device="cuda")
    gd, xd, wd = g.double(), x.double(), w.double()
    for epi, aux in ((_C.EPI_NONE, None), (_C.EPI_DSILU, pre)):
        dW0 = torch.ones(no, ni, device="cuda")
        db0 = torch.ones(no, device="cuda")
        dW, dX, db = _C.linear_bwd(g, x, w, True, epi, aux, dW=dW0, accumulate=True, bias=True, db=db0, accumulate_db=True)
        want_dW = 1.0 + gd.T @ xd
        want_dX = gd @ wd
        if aux is not None:
            s = torch.sigmoid(pre.double())
            want_dX = want_dX * (s * (1 + pre.double() * (1 - s)))
        e1 = float((dW.double() - want_dW).abs().max() / want_dW.abs().max())
        e2 = float((dX.double() - want_dX).abs().max() / want_dX.abs().max())
        e3 = float((db.double() - (1.0 + gd.sum(0))).abs().max() / gd.sum(0).abs().max())
        ok = e1 < 2e-6 and e2 < 2e-6 and e3 < 2e-6
        bad += not ok
        print(f"B={B} {no}x{ni} epi={epi}: dW {e1:.2e} dX {e2:.2e} db {e3:.2e} {'ok' if ok else 'BAD'}", flush=True)
    dW, dX = _C.linear_bwd(g, x, w, False)
    e1 = float((dW.double() - gd.T @ xd).abs().max() / (gd.T @ xd).abs().max())
    d1, _ = _C.linear_bwd(g, x, w, False)
    for _ in range(3):
        d2, dX2, db2 = _C.linear_bwd(g, x, w, True, bias=True)
        d3, dX3, db3 = _C.linear_bwd(g, x, w, True, bias=True)
        assert torch.equal(d2, d3) and torch.equal(dX2, dX3) and torch.equal(db2, db3), 'not deterministic'
    det = torch.equal(dW, d1)
    bad += not (e1 < 2e-6 and det and dX is None)
    print(f"   no-dX: dW {e1:.2e} deterministic {det}")
print("BAD" if bad else "ALL OK")
if bad:
    sys.exit(1)
for B, no, ni, dx in [(1024, 768, 512, 1), (1024, 512, 256, 1), (1024, 256, 128, 1), (1024, 128, 256, 1), (1024, 256, 512, 1), (1024, 512, 768, 0),
                      (1024, 691, 768, 1), (1024, 768, 691, 1), (1024, 460, 512, 1), (1024, 512, 460, 1), (1024, 230, 256, 1), (1024, 345, 691, 1),
                      (2048, 768, 512, 1)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    dW = torch.empty(no, ni, device="cuda"); db = torch.empty(no, device="cuda")
    t = bench.time_kernel(lambda: _C.linear_bwd(g, x, w, need_dx=bool(dx), dW=dW, bias=True, db=db))
    fl = 2.0 * B * no * ni * (2 if dx else 1)
    print(f"MID={os.environ.get('HIDVAE_GEMM_MID')} PF={os.environ.get('HIDVAE_GEMM_MID_PF', '2')} lbwd B={B} {no}x{ni} dx={dx}: {t:6.1f} us  {fl / t * 1e-6:6.1f} TFLOP/s", flush=True)
