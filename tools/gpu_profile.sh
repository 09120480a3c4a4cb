#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of bench.py. Usage: tools/gpu_profile.sh <tag>
# Counters are collected in their own runs (no --sys-trace etc. together with --pmc).
# K1=tall tools/gpu_profile.sh <tag>: every launch takes the threshold kernel's tall-strip instance (k_threshold<2, false, 7>, what the pipelined loop
# launches for large batches) instead of the short-strip one a call that runs alone gets -- the --no-pipeline traces then time THAT kernel alone.
TAG=${1:-run}
R=$PWD
if [ -n "$K1" ]; then export CIMBAR_HIP_K1_STRIPS=$K1; fi
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# kernel-level evidence is taken with --no-pipeline: every dispatch then runs alone, which is also how bench.py itself measures
# stage_ms / roofline (its timing reps are ordinary calls); the default (pipelined) command is traced as well for the record
BENCH="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pipeline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log | tail -1 > ${OUT}_bench_under_trace.json
mkdir -p $OUT/tracep
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tracep -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/tracep.log 2>&1
grep '"metric"' $OUT/tracep.log | tail -1 > ${OUT}_bench_under_trace_pipelined.json
python - "$OUT/tracep" "${OUT}_kernel_stats_pipelined.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        r = csv.reader(f)
        head = next(r)
        rows = [head] + [x for x in r if "k_threshold" in x[0] or "::k_" in x[0]]
with open(sys.argv[2], "w", newline="") as f:
    csv.writer(f).writerows(rows)
PY
rm -rf $OUT/tracep
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1; head -c 3000 $OUT.log; rm -rf $OUT
