#!/bin/bash
# Runs on the GPU box (via gpurun): HBM traffic counters (FETCH_SIZE, WRITE_SIZE, TCC hits / misses) of the configs[4] kernels.  Usage: <tag> <captures>
TAG=${1:-run}; N=${2:-1024}
R=$PWD
OUT=$R/gpurun_out/c5mem_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/dense_probe.py --worker $N 1"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_1 -o p -- $CMD > $OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_2 -o p -- $CMD > $OUT/pmc2.log 2>&1
python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1
rm -rf $OUT/pmc_*
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "_summary.json"))["pmc"]
for k, v in d.items():
    if "FETCH_SIZE" in v:
        print(k[:60], {a: v[a] for a in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum") if a in v}, v.get("_grid"))
PY
