#!/usr/bin/env python3
"""Per-kernel-family matrix-core occupancy from one rocprofv3 --pmc pass over bench.py (eager launches):

    python tools/pmc_mfma.py <pmc_dir> <out.json>

Counters (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_VALU_MFMA_BUSY_CYCLES counts shader CYCLES in which a SIMD's
matrix pipe is busy, summed over the chip; SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE give the kernel's duration in cycles.
mfma_busy = MFMA_BUSY / (1024 SIMDs x duration cycles).  Only the LAST training step's dispatches are used.
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


def main():
    d, out = sys.argv[1:3]
    rows = defaultdict(dict)
    order = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            rows[k][r["Counter_Name"]] = float(r["Counter_Value"])
            order[k] = (family(r["Kernel_Name"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ids = sorted(rows)
    marks = [i for i in ids if order[i][0] == "adam_kernel"]
    if len(marks) >= 2:
        ids = [i for i in ids if marks[-2] < i <= marks[-1]]
    acc = defaultdict(lambda: defaultdict(float))
    for i in ids:
        fam, ns = order[i]
        a = acc[fam]
        a["launches"] += 1
        a["ns"] += ns
        for c, v in rows[i].items():
            a[c] += v
    res = {}
    for fam, a in acc.items():
        if a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0        # summed over the 8 XCDs
        e = {"launches": int(a["launches"]), "avg_ns": a["ns"] / a["launches"],
             "mfma_busy_cycles_per_launch": a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["launches"],
             "gui_active_cycles_per_launch": cyc / a["launches"]}
        # Denominator: the kernel's duration x the NOMINAL shader clock (2.4 GHz, MI355X_MICROARCH.md) x 1024 SIMDs.  The
        # GRBM_GUI_ACTIVE / 8 quotient reads high on dispatches shorter than ~0.3 ms (the guide's DVFS note; round 2's table
        # showed an "effective clock" of 2.83 GHz, above the chip's maximum -- ADVICE r2), so it is kept only as a
        # cross-check; at the true (lower) clock under load the busy fraction is somewhat HIGHER than this figure.
        e["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * a["ns"] * 2.4)
        e["clock_note"] = "busy cycles / (duration_ns x 2.4 GHz nominal x 1024 SIMDs): a lower bound of the busy fraction"
        if cyc > 0:
            e["mfma_busy_frac_gui_active"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
            e["gui_active_clock_ghz"] = cyc / a["ns"]
            e["gui_active_clock_plausible"] = bool(cyc / a["ns"] <= 2.4 * 1.02)
        for c in ("SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_MOPS_F32"):
            if a.get(c):
                e[c.lower() + "_per_launch"] = a[c] / a["launches"]
        res[fam] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:8]:
        print(f"{k:34s} n={v['launches']:4d} avg {v['avg_ns'] / 1e3:7.1f} us  mfma busy {v.get('mfma_busy_frac', float('nan')):.3f}"
              f"  (GUI_ACTIVE clock {v.get('gui_active_clock_ghz', float('nan')):.2f} GHz)")


if __name__ == "__main__":
    main()
