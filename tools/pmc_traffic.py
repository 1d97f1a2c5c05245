#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into per-kernel-family HBM traffic.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md section HBM).  On gfx950 FETCH_SIZE under-reports wide
coalesced reads by 2x; the factor for this code's access widths is CALIBRATED in the same run on `channel_sum_kernel`
(4-byte-per-lane coalesced reads of an exactly known byte count) and applied to the conv kernels, which also read
4 bytes per lane.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


def collect(d, counter):
    """Per-family counter values of the LAST training step only (dispatches between the last two adam_kernel
    launches), so the autotuner's trial launches of the first step do not pollute the averages."""
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), family(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[1] == "adam_kernel"]
    if len(marks) >= 2:
        rows = rows[marks[-2] + 1: marks[-1] + 1]
    acc = defaultdict(list)
    for _, fam, v in rows:
        acc[fam].append((v, None))
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, write = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fv = [v for v, _ in fetch.get(k, [])]
        wv = [v for v, _ in write.get(k, [])]
        res[k] = {"launches": max(len(fv), len(wv)),
                  "fetch_kib_per_launch": sum(fv) / max(1, len(fv)),
                  "write_kib_per_launch": sum(wv) / max(1, len(wv))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["fetch_kib_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:34s} n={v['launches']:5d} fetch {v['fetch_kib_per_launch'] / 1024:9.2f} MiB  write {v['write_kib_per_launch'] / 1024:9.2f} MiB per launch")


if __name__ == "__main__":
    main()
