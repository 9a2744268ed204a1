"""A/B of the co-resident form of hidvae_linear_bwd at a given batch: runs bench.py's tagged workload with and without announcing the
level streams (python scratch/r3/co_resident_ab.py --batch 4096 --levels 4 --codes 1024)"""
import subprocess, sys, os
args = sys.argv[1:]
for off in (0, 1):
    code = ("import sys; sys.argv=['bench.py','--cpu-seconds','0','--kernels','0','--also-large','0','--also-other','0','--steps','40'] + %r\n"
            "import hidvae_amd\nfrom hidvae_amd import _C\n" % (args,))
    if off:
        code += "_C.register_ws_lane = lambda s: int(s.cuda_stream)\n"
    code += "import bench; bench.main()\n"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print("co_resident", "off" if off else "on ", out.stdout.strip()[-90:], flush=True)
