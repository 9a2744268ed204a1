#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) of
tools/profile_kernels.py into the evidence file bench.py reads:
    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_quoted_kernels_pmc_hbm_traffic.csv
Values are RAW counter averages per launch in KB (on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads;
bench.py doubles it, the file does not)."""
import csv
import glob
import re
import sys
from collections import defaultdict

# bench.py kernel-name prefix  <-  (substring of the profiled kernel name, grid size in work-items)
QUOTED = [
    ("gemm_tile16_kernel encoder layer 0", "gemm_tile16_kernel<2, 4>", None),
    ("rq_forward_kernel (fused L-level VQ, code-split variant)", "rq_forward_kernel<3, true, true, true", None),
    ("rq_forward at 1,048,576 items", "rq_forward_pf32_kernel<3, true", None),
    ("gemm_f32_kernel<2,2,NT> encoder layer 0 at 65,536 rows", "gemm_f32_kernel<2, 2, 0>", None),
    # four launches of the ring kernel, told apart by their grids (workgroups x 256 threads; gemm_ring.hip: G = min(512, S / 8) ranges)
    ("gemm_ring_bwd_kernel encoder layer 0 backward, weight gradient only", "gemm_ring_bwd_kernel", 101376),  # 396 ranges of 8 steps
    ("gemm_ring_bwd_kernel decoder layer 3 backward", "gemm_ring_bwd_kernel", 131072),                        # 1024 x 768 x 512, no bias
    ("gemm_ring_bwd_kernel tag-head layer backward 691 x 768 at B = 1024", "gemm_ring_bwd_kernel", 129280),   # 505 ranges of 17 steps
    ("gemm_ring_bwd_kernel tag-head layer backward 691 x 768 at B = 2048", "gemm_ring_bwd_kernel", 128768),   # 503 ranges of 34 steps
    ("rq_forward_kernel streamed code-split (4x1024, B=4096)", "rq_forward_kernel<3, true, false, true, 4, true>", None),
    ("gather_rows_kernel", "gather_rows_kernel", None),
    ("rq_forward ids-only at 1,048,576 items", "rq_forward_pf32_kernel<2, false, 16, 8, true, true>", None),
]


def averages(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        acc[(name, r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, n = averages(sys.argv[1], "FETCH_SIZE")
    write, _ = averages(sys.argv[2], "WRITE_SIZE")
    w = csv.writer(sys.stdout)
    w.writerow(["bench_kernel_prefix", "kernel", "grid_work_items", "launches", "FETCH_SIZE_KB", "WRITE_SIZE_KB"])
    for prefix, sub, _ in QUOTED:
        keys = [k for k in fetch if sub in k[0] and (_ is None or int(k[1]) == _)]
        if not keys:
            continue
        k = max(keys, key=lambda kk: int(kk[1]))  # the quoted launch is the largest grid of that instantiation
        w.writerow([prefix, k[0], k[1], n[k], f"{fetch[k]:.1f}", f"{write.get(k, 0.0):.1f}"])


if __name__ == "__main__":
    main()
