#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass (counter_collection.csv) into per-kernel averages per launch:
    python tools/pmc_table.py <dir of the pass> [substring ...] > profiles/rNN_....csv
Rows: kernel (anonymous-namespace prefix stripped), grid size, launches, then one column per counter (average per launch)."""
import csv
import glob
import re
import sys
from collections import defaultdict


def main():
    f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
    want = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    counters = []
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if want and not any(w in name for w in want):
            continue
        if r["Counter_Name"] not in counters:
            counters.append(r["Counter_Name"])
        acc[(name, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "grid_work_items", "launches"] + counters)
    for (name, grid), cs in sorted(acc.items(), key=lambda kv: -sum(kv[1][counters[0]])):
        n = len(cs[counters[0]])
        w.writerow([name[:150], grid, n] + [f"{sum(cs[c]) / max(1, len(cs[c])):.1f}" for c in counters])


if __name__ == "__main__":
    main()
