"""hand-over stress for gemm_mid_sk_kernel (even ranges: partial tiles cross workgroups through memory): thousands of launches on a few
shapes, every result compared bit for bit with the first; other traffic (a big copy) running on a second stream to disturb the caches"""
import os, sys, time
sys.path.insert(0, ".")
import torch
import hidvae_amd
from hidvae_amd import _C
torch.manual_seed(1)
N = int(os.environ.get("N", "3000"))
side = torch.cuda.Stream()
big_a, big_b = torch.randn(64 << 20, device="cuda"), torch.empty(64 << 20, device="cuda")
bad = 0
for B, no, ni in [(2048, 768, 512), (1024, 691, 768), (4096, 512, 256), (8192, 256, 128), (3000, 333, 385)]:
    g = torch.randn(B, no, device="cuda"); x = torch.randn(B, ni, device="cuda"); w = torch.randn(no, ni, device="cuda")
    ref = [t.clone() for t in _C.linear_bwd(g, x, w, True, bias=True)]
    gd = g.double()
    assert float((ref[0].double() - gd.T @ x.double()).abs().max() / (gd.T @ x.double()).abs().max()) < 2e-6
    t0 = time.time()
    for it in range(N):
        if it % 50 == 0:
            with torch.cuda.stream(side):
                big_b.copy_(big_a)
        out = _C.linear_bwd(g, x, w, True, bias=True)
        if it % 10 == 0 or it > N - 20:
            if not all(torch.equal(a, b) for a, b in zip(out, ref)):
                bad += 1
                print(f"MISMATCH B={B} {no}x{ni} launch {it}: {[float((a - b).abs().max()) for a, b in zip(out, ref)]}", flush=True)
    torch.cuda.synchronize()
    print(f"B={B} {no}x{ni}: {N} launches, {bad} mismatches so far, {time.time() - t0:.1f} s", flush=True)
print("STRESS", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
