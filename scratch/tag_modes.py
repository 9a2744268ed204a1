"""tagged step time: grouped lockstep (one stream) vs per-level branches on 3 streams, three configurations"""
import argparse, os, sys, subprocess
sys.path.insert(0, ".")
import torch
import bench
for B, L, K in ((1024, 3, 256), (2048, 3, 256), (4096, 4, 1024)):
    for grouped in ("1", "0"):
        os.environ["HIDVAE_TAG_GROUPED"] = grouped
        a = argparse.Namespace(gpus=1, steps=40, warmup=5, batch=B, levels=L, codes=K, tagged=1, graph=1, cpu_seconds=0, pool=2, dist=0, kernels=0,
                               also_large=0, also_tagged=0, windows=3)
        dt, _, info = bench.run_workload(a, torch.device("cuda", 0), 0, 1, None)
        print(f"B={B} {L}x{K} grouped={grouped}: {dt / a.steps * 1e3:.4f} ms/step  windows {['%.4f' % w for w in info['windows_ms_per_step']]}", flush=True)
        del info
        torch.cuda.empty_cache()
