#!/bin/bash
# quick: one PMC pass (instruction mix + cycles) of bench.py, our kernels only. Usage: tools/gpu_quick_pmc.sh <tag>
TAG=${1:-q}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1; python - <<PY
import json
d=json.load(open("$OUT" + "_summary.json"))
for k,v in d["pmc"].items(): print(k, {a:b for a,b in v.items() if not a.startswith("_")})
PY
rm -rf $OUT
