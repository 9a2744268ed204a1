"""Phase timestamps (s_memtime) of one workgroup of the fused middle launch; needs the instrumented library scratch/_lib_stamps.so
copied over hid-vae_amd/libhidvae_hip.so (built from csrc/rq.hip with STAMP() calls, see the session notes in DESIGN.md)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hidvae_amd
from hidvae_amd import _C
dev = torch.device("cuda"); torch.manual_seed(0)
B, K2, N2, Nd0, Nd1, L, K = 1024, 256, 128, 128, 256, 3, 256
h1 = torch.randn(B, K2, device=dev); W2 = torch.randn(N2, K2, device=dev) * 0.06; W3 = torch.randn(32, N2, device=dev) * 0.09
Wd0 = torch.randn(Nd0, 32, device=dev) * 0.2; Wd1 = torch.randn(Nd1, Nd0, device=dev) * 0.09
tabs = [(torch.rand(K, 32, device=dev) * 2 - 1) * (1.0 if i == 0 else 0.35 * 0.5 ** i) for i in range(L)]
cb, cc = _C.codebook_prepare(tabs, [i == 0 for i in range(L)])
SCR = _C.census_scratch(B, dev)
names = ["start", "staged", "enc2 done", "enc3 done", "levels done", "dec0 done", "dec1 done", "end"]
lib = _C.lib()
for rep in range(4):
    _C.bottleneck_fwd(h1, W2, W3, cb, cc, True, 3, 0.4, Wd0, Wd1, id_stats=True, scratch=SCR)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert lib.hidvae_debug_stamps(out) == 0
    t = [out[i] for i in range(8)]
    print("rep", rep, " ".join(f"{names[i]}:+{(t[i]-t[i-1])}" for i in range(1, 8)), "total (10 ns ticks)", t[7] - t[0], flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): _C.bottleneck_fwd(h1, W2, W3, cb, cc, True, 3, 0.4, Wd0, Wd1, id_stats=True, scratch=SCR)
e1.record(); torch.cuda.synchronize()
print("launch-to-launch us:", e0.elapsed_time(e1) * 50)
