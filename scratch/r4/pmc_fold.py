"""fold rocprofv3 counter_collection.csv files under a directory: per kernel (name prefix), per counter: mean over launches"""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if not any(k in name for k in ("gemm_", "ring")):
            continue
        acc[(name[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(acc.items()):
    v = v[2:] if len(v) > 4 else v
    print(f"{name:62s} {c:28s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("gemm_", "ring")):
            print("stats:", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"])
