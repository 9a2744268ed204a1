import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "_store_stream.so")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "store_stream.hip")])
lib = ctypes.CDLL(so)
dev = torch.device("cuda")
B = 1 << 20
y = torch.randn(B, 32, device=dev); z = torch.empty(B, 32, device=dev); cat = torch.empty(B, 96, device=dev); es = torch.empty(B, 32, device=dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for grid in (256, 512):
  for spin in (0, 200, 1000):
    for var in (0, 1, 2):
        def run():
            rc = lib.run_stream(var, p(y), ctypes.c_int64(B), p(z), p(cat), p(es), spin, grid, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"grid {grid} spin {spin:5d} var {var}: {us:8.1f} us   {B * 768 / us * 1e-6:6.2f} TB/s moved", flush=True)
