#!/usr/bin/env python3
"""Per-launch timeline of one captured train step, taken with device timestamps around every C-ABI launch (_C.LaunchStamps).

    python tools/step_timeline.py [--tagged 1] [--batch 1024] [--levels 3] [--codes 256] [--top 40]

Prints, per entry point: launches, summed in-step microseconds (launch + one dependent-launch gap), share of the sum; then the
un-stamped step time for reference.  The stamped replay is slower than the real step (two extra launches per launch): only the
per-launch brackets are meaningful, not its total."""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def timeline(args, device):
    a = argparse.Namespace(**{**vars(args), "steps": 20, "warmup": 5, "graph": 1, "dist": 0, "windows": 1})
    _, _, info = bench.run_workload(a, device, 0, 1, None)
    rows, empty = bench.step_timeline(info["stepper"], info["pool_batch"], device)
    return [(r["entry"] + (f" {r['M']}x{r['N']}x{r['K']}" if "M" in r else ""), r["us"]) for r in rows], empty


def plain_step_ms(args, device, steps=100):
    a = argparse.Namespace(**{**vars(args), "steps": steps, "warmup": 10, "graph": 1, "dist": 0, "windows": 3})
    dt, _, _ = bench.run_workload(a, device, 0, 1, None)
    return dt / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tagged", type=int, default=0)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--each", type=int, default=0, help="also print every launch in order")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    rows, empty = timeline(args, device)
    agg = {}
    for n, v in rows:
        c = agg.setdefault(n, [0, 0.0, 0.0])
        c[0] += 1
        c[1] += v
        c[2] = max(c[2], v)
    total = sum(v for _, v in rows)
    print(f"# {'tagged' if args.tagged else 'untagged'} step, B={args.batch}, {args.levels}x{args.codes}: {len(rows)} launches, "
          f"sum of brackets {total:.1f} us (empty bracket {empty:.2f} us subtracted from each)")
    print(f"{'entry point':48s} {'n':>4s} {'sum us':>9s} {'max us':>8s} {'share':>6s}")
    for n, (c, s, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: args.top]:
        print(f"{n:48s} {c:4d} {s:9.1f} {mx:8.1f} {100 * s / total:5.1f}%")
    if args.each:
        for i, (n, v) in enumerate(rows):
            print(f"{i:4d} {n:48s} {v:8.2f}")
    print(f"# un-stamped replayed step: {plain_step_ms(args, device):.4f} ms")


if __name__ == "__main__":
    main()
