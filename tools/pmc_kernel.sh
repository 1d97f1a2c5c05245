#!/bin/bash
# SQ counters of one kernel shape (two rocprofv3 --pmc passes, counters only):
#   bash tools/pmc_kernel.sh <tag> <one_conv.py arguments...>
# Prints per-dispatch averages of the named kernel family; raw CSVs under gpurun_out/<tag>/.
set -o pipefail
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $out/p1 -- python3 tools/one_conv.py "$@" > $out/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/p2 -- python3 tools/one_conv.py "$@" > $out/p2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st -- python3 tools/one_conv.py "$@" > $out/st.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "conv3x3" not in k and "wgrad" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[len(v) // 2:]          # skip warm-up dispatches
        print(f"   {c:32s} {sum(v) / len(v):16.0f}")
for f in glob.glob(out + "/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Name"] or "wgrad" in r["Name"]:
            print("   avg ns", r.get("AverageNs"), "calls", r.get("Calls"), r["Name"][:70])
PY
rm -rf $out/st
