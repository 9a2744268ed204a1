#!/usr/bin/env python3
"""Which launches of the captured train step run beside each other: one lane per stream, absolute device timestamps.

    python tools/step_gantt.py [--tagged 1] [--batch 1024] [--levels 3] [--codes 256] [--each 1]

Per stream: launches, busy time, first start / last end, idle gaps; then (with --each) every launch with its lane, start, end.
The stamped replay carries two timestamp launches per launch, so absolute times are stretched; what matters is the STRUCTURE --
which stream is the long pole, how much of a lane is idle -- next to tools/step_timeline.py's per-launch durations."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tagged", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--levels", type=int, default=3)
    ap.add_argument("--codes", type=int, default=256)
    ap.add_argument("--pool", type=int, default=4)
    ap.add_argument("--each", type=int, default=0)
    args = ap.parse_args()
    from hidvae_amd import _C
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    a = argparse.Namespace(**{**vars(args), "steps": 20, "warmup": 5, "graph": 1, "dist": 0, "windows": 1})
    _, _, info = bench.run_workload(a, device, 0, 1, None)
    stepper, pool_batch = info["stepper"], info["pool_batch"]
    inner = stepper._fwd_bwd
    graphs = stepper.graphs
    stepper.graphs = None
    stamps = _C.stamps_begin(device)
    try:
        stepper([pool_batch(0)])
    finally:
        _C.stamps_end()
    for i in range(3):
        stepper([pool_batch(1 + i)])
    torch.cuda.synchronize()
    spans = stamps.spans()
    stepper.graphs = graphs
    lanes = {}
    for n, s, t0, t1 in spans:
        lanes.setdefault(s, []).append((t0, t1, n))
    order = sorted(lanes, key=lambda s: min(t for t, _, _ in lanes[s]))
    end = max(t1 for _, _, _, t1 in spans)
    print(f"# {'tagged' if args.tagged else 'untagged'} step B={args.batch} {args.levels}x{args.codes}: {len(spans)} launches on {len(lanes)} streams, "
          f"stamped span {end:.1f} us")
    for li, s in enumerate(order):
        ev = sorted(lanes[s])
        busy = sum(b - a_ for a_, b, _ in ev)
        gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
        big = sorted(((g, ev[i][2], ev[i + 1][2]) for i, g in enumerate(gaps) if g > 15.0), reverse=True)[:6]
        print(f"lane {li} (stream {s:#x}): {len(ev)} launches, busy {busy:.1f} us, from {ev[0][0]:.1f} to {ev[-1][1]:.1f} us, "
              f"idle inside {sum(g for g in gaps if g > 0):.1f} us")
        for g, before, after in big:
            print(f"      idle {g:7.1f} us between {before} and {after}")
    if args.each:
        idx = {s: i for i, s in enumerate(order)}
        for n, s, t0, t1 in sorted(spans, key=lambda r: r[2]):
            print(f"{idx[s]} {t0:9.1f} {t1:9.1f} {t1 - t0:7.1f} {n}")


if __name__ == "__main__":
    main()
