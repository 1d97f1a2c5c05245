#!/usr/bin/env python3
"""Aggregate a rocprofv3 --stats kernel_stats CSV by kernel family (template instantiations merged):
    python tools/stats_by_family.py profiles/r01/rocprofv3_kernel_stats.csv
The per-family average is what bench.py's roofline.avg_launch_us (HIP events, live) should agree with."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
fam = collections.defaultdict(lambda: [0, 0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    f = m.group(1) if m else name
    fam[f][0] += int(r["Calls"])
    fam[f][1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in fam.values())
print(f"{'kernel family':36s} {'calls':>6s} {'avg us':>9s} {'share':>7s}")
for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{f:36s} {c:6d} {t / c / 1000:9.2f} {100 * t / tot:6.1f}%")
