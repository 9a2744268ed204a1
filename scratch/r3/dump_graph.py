"""dump the captured tagged step graph (hipGraphDebugDotPrint) to look at the edges between the level branches"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
orig = torch.cuda.CUDAGraph
made = []
class G(orig):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        return g
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.enable_debug_mode()
        made.append(self)
torch.cuda.CUDAGraph = G
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ap = bench.parse()
a = argparse.Namespace(**{**vars(ap), "tagged": 1, "steps": 5, "warmup": 2, "graph": 1, "dist": 0, "windows": 1})
bench.run_workload(a, dev, 0, 1, None)
out = os.path.join("gpurun_out", "r3b_tagged_graph.dot")
made[0].debug_dump(out)
print("dumped", out, os.path.getsize(out))
