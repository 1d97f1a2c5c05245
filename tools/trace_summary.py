#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the LAST `--steps` graph replays of bench.py and the
per-launch list of one step (so each launch can be matched to its layer).

    python tools/trace_summary.py gpurun_out/prof/.../*_kernel_trace.csv [--one-step]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                         int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", 0) or 0)))
    rows.sort()
    # a step = the launches between two consecutive adam_kernel launches
    idx = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
    if len(idx) < 3:
        print("not enough steps in trace"); return
    a, b = idx[-2] + 1, idx[-1] + 1
    step = rows[a:b]
    wall = (step[-1][1] - step[0][0]) / 1e3
    busy = sum(r[1] - r[0] for r in step) / 1e3
    print(f"one step: {len(step)} launches, wall {wall:.1f} us, sum of kernel durations {busy:.1f} us")
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, n, g, w in step:
        tot[n][0] += 1
        tot[n][1] += (e - s) / 1e3
    for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:9.1f} us  {c:4d}x  {n}")
    if "--one-step" in sys.argv:
        print("---- launches in order ----")
        t0 = step[0][0]
        for s, e, n, g, w in step:
            print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  grid {g // max(w, 1):6d} x {w:4d}  {n}")


if __name__ == "__main__":
    main()
