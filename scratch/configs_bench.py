"""The BASELINE.json configurations that are not the bench line (configs 3 and 5) + GPU k-means start-up time."""
import json, os, subprocess, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
def bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-seconds", "0", "--kernels", "0", "--also-large", "0",
                          "--also-tagged", "0", "--steps", "100", "--warmup", "10", "--pool", "2", *flags], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else out.stderr[-400:]
print("config 3 (B=2048, tagged):", bench("--batch", "2048", "--tagged", "1"), flush=True)
print("config 5 (B=4096, 4x1024, untagged):", bench("--batch", "4096", "--levels", "4", "--codes", "1024"), flush=True)
print("config 5 tagged (B=4096, 4x1024):", bench("--batch", "4096", "--levels", "4", "--codes", "1024", "--tagged", "1"), flush=True)
import torch, hidvae_amd
from hidvae_amd.init.kmeans import Kmeans
torch.manual_seed(0)
for K in (256, 1024):
    x = torch.nn.functional.normalize(torch.randn(20000, 32, device="cuda"), dim=-1)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        out = Kmeans(k=K, max_iters=None).run(x)
        torch.cuda.synchronize(); dt = time.time() - t0
    print(f"k-means 20000 x 32 -> K={K}: {dt*1e3:.1f} ms", flush=True)
