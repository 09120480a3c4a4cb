#!/bin/bash
# GPU: dynamic instruction counts of mode 66's threshold kernel in the library as built and in a variant library (CIMBAR_HIP_LIB), one PMC pass each
# over tools/mode_step_probe.py 66. Usage: tools/gpu_pmc_mode66.sh <variant.so>  ->  gpurun_out/pmc_mode66.txt
R=$PWD
OUT=$R/gpurun_out/pmc66
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in default variant; do
	if [ $v = variant ]; then export CIMBAR_HIP_LIB=$R/$1; fi
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/$v -o pmc -- python $R/tools/mode_step_probe.py 66 > $OUT/$v.log 2>&1
done
python - <<PY > $R/gpurun_out/pmc_mode66.txt
import csv, glob, collections
csv.field_size_limit(1 << 30)
for v in ("default", "variant"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_threshold" not in k: continue
            k = k[k.index("k_threshold"):].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen: seen.add(key); n[k] += 1
    for k in acc:
        print(v, k, "launches", n[k], {c: round(x / n[k]) for c, x in sorted(acc[k].items())})
PY
rm -rf $OUT
