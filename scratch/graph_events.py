"""experiment: can timing events be recorded INSIDE a captured HIP graph (torch.cuda.Event(external=True)) and read after replay?"""
import torch, sys
sys.path.insert(0, ".")
import hidvae_amd
from hidvae_amd import _C
x = torch.randn(1024, 768, device="cuda"); w = torch.randn(512, 768, device="cuda"); o = torch.empty(1024, 512, device="cuda"); a = torch.empty_like(o)
f = lambda: _C.gemm(_C.GEMM_NT, x, w, out=o, epilogue=_C.EPI_SILU, aux=a)
f(); torch.cuda.synchronize()
for ext in (True, False):
    try:
        kw = dict(enable_timing=True)
        if ext:
            kw["external"] = True
        e0, e1, e2 = (torch.cuda.Event(**kw) for _ in range(3))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            f(); e0.record(); f(); e1.record(); f(); f(); e2.record()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print("external" if ext else "plain", "one launch us:", e0.elapsed_time(e1) * 1e3, "two launches us:", e1.elapsed_time(e2) * 1e3)
    except Exception as e:
        print("external" if ext else "plain", "FAILED:", type(e).__name__, str(e)[:300])
