#!/bin/bash
# Round-end measurement batch on the GPU box (run through gpurun from the repo root):
#   bash tools/measure_round.sh r01
# Writes everything under gpurun_out/<tag>/; copy what should be judged into profiles/<tag>/.
#   1. pytest -m gpu log          2. bench.py JSON line (with the CPU baseline leg)
#   3. rocprofv3 --kernel-trace --stats of the same bench command (kernel_stats CSV)
#   4. two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only, no trace domains) -> pmc_traffic.json
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > $out/pytest_gpu.log 2>&1; tail -2 $out/pytest_gpu.log
export CM_TUNE_CACHE=$out/tuned.txt   # written by the first run; the profiled runs load it and do no tuning launches
python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python tools/show_bench.py $out/bench.json > $out/bench_table.txt; head -8 $out/bench_table.txt
#   2b. secondary line: BASELINE configs[3] (cnn_transformer, training mode with dropout)
python bench.py --model cnn_transformer --steps 30 --warmup 5 > $out/bench_cnn_transformer.json 2> $out/bench_cnn_transformer.err && python tools/show_bench.py $out/bench_cnn_transformer.json > $out/bench_cnn_transformer_table.txt && head -6 $out/bench_cnn_transformer_table.txt
#   2c. one rank's share of BASELINE configs 3 and 5 (set CM_MEASURE_BIG=0 to skip: ~2.5 minutes)
if [ "${CM_MEASURE_BIG:-1}" != "0" ]; then
  python bench.py --base 64 --seq-len 12 --steps 10 --warmup 3 > $out/bench_config3_one_rank.json 2> $out/bench_config3.err
  python bench.py --base 64 --height 192 --width 288 --batch 16 --steps 5 --warmup 2 > $out/bench_config5_one_rank.json 2> $out/bench_config5.err
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > $out/stats.log 2>&1
f=$(ls $out/stats/*/*kernel_stats.csv 2>/dev/null | tail -1); [ -n "$f" ] && cp $f $out/rocprofv3_kernel_stats.csv && python tools/stats_by_family.py $out/rocprofv3_kernel_stats.csv > $out/rocprofv3_kernel_stats_by_family.txt
t=$(ls $out/stats/*/*kernel_trace.csv 2>/dev/null | tail -1); [ -n "$t" ] && python tools/trace_summary.py $t --one-step > $out/one_step_trace.txt
rm -rf $out/stats
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph > $out/pmc_write.log 2>&1
python tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json | head -8
rm -rf $out/pmc_fetch $out/pmc_write
#   5. matrix-core occupancy (north_star's "MFMA-busy counters"): SQ + GRBM counters, counters only
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-graph > $out/pmc_mfma.log 2>&1
python tools/pmc_mfma.py $out/pmc_mfma $out/pmc_mfma.json | head -8
rm -rf $out/pmc_mfma
ls -la $out
