#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of BASELINE configs[4] at one batch size through tools/dense_probe.py's worker
# (device-resident 1080p captures -> scan, extract, decode; every frame takes the exact replay).  Usage: tools/gpu_profile_config5.sh <tag> <captures> [pmc]
TAG=${1:-run}
N=${2:-2048}
R=$PWD
OUT=$R/gpurun_out/config5_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/dense_probe.py --worker $N 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python - "$OUT/trace" "${OUT}_kernel_stats.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        r = csv.reader(f)
        head = next(r)
        rows = [head] + [x for x in r if "m68" in x[0]]
with open(sys.argv[2], "w", newline="") as f:
    csv.writer(f).writerows(rows)
for r in rows[:40]:
    print(r[0].replace("(anonymous namespace)::", "")[:60], r[1:4])
PY
if [ -n "$3" ]; then
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1
fi
rm -rf $OUT/trace
