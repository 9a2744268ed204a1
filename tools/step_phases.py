#!/usr/bin/env python3
"""Where the replayed train step is at each phase boundary: ~30 single device timestamps (hidvae_amd._C.PhaseMarks) instead of
a bracket around every launch, so the streams keep their real timing against each other.

    python tools/step_phases.py [--tagged 1] [--batch 1024] [--levels 3] [--codes 256] [--replays 20]

Prints every mark's median time (us after the step's first mark) over the replays, with the stream it was taken on."""
import argparse
import os
import statistics
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
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--ahead", type=int, default=6, help="back-to-back replays per sample (1 = every replay starts on an idle GPU)")
    args = ap.parse_args()
    from hidvae_amd import _C
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    a = argparse.Namespace(**{**vars(args), "steps": 20, "warmup": 5, "graph": 1, "dist": 0, "windows": 1})
    dt, _, info = bench.run_workload(a, device, 0, 1, None)
    stepper, pool_batch = info["stepper"], info["pool_batch"]
    stepper.graphs = None
    marks = _C.phases_begin(device)
    try:
        stepper([pool_batch(0)])  # re-captures the step with the marks in it
    finally:
        _C.phases_end()
    runs = []
    for i in range(args.replays):
        for k in range(args.ahead):  # back-to-back replays: the marks of the LAST one show the steady state, in which the host has
            stepper([pool_batch(1 + i + k)])  # queued a step's nodes while the previous step was still running
        torch.cuda.synchronize()
        runs.append(marks.rows())
    streams = {}
    for _, st, _ in runs[0]:
        streams.setdefault(st, len(streams))
    print(f"# {'tagged' if args.tagged else 'untagged'} step B={args.batch} {args.levels}x{args.codes}: un-marked step {dt / a.steps * 1e3:.4f} ms; "
          f"{len(runs[0])} marks, median over {len(runs)} replays")
    med = [(lab, streams[st], statistics.median(r[i][2] for r in runs)) for i, (lab, st, _) in enumerate(runs[0])]
    for lab, lane, t in sorted(med, key=lambda r: r[2]):
        print(f"{t:9.1f} us  lane {lane}  {lab}")


if __name__ == "__main__":
    main()
