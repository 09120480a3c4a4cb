#!/bin/bash
# GPU: LDS counters of the threshold kernel in the library as built and in a variant library, one PMC pass each over tools/mode_step_probe.py <mode>.
# Usage: tools/gpu_pmc_k1_lds.sh <variant.so> [mode]  ->  gpurun_out/pmc_k1_lds.txt
R=$PWD
MODE=${2:-68}
OUT=$R/gpurun_out/pmclds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default variant; do
	if [ $v = variant ]; then export CIMBAR_HIP_LIB=$R/$1; fi
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -o pmc -- python $R/tools/mode_step_probe.py $MODE > $OUT/$v.log 2>&1
done
python - <<PY > $R/gpurun_out/pmc_k1_lds.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
for v in ("default", "variant"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_threshold" not in k and "k_rs<" not in k: continue
            k = k[k.index("k_"):].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[k] += 1
    for k in sorted(acc):
        print(v, k, "launches", n[k], {c: round(x / n[k]) for c, x in sorted(acc[k].items())})
PY
rm -rf $OUT
