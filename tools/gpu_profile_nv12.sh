#!/bin/bash
# GPU: rocprofv3 kernel trace of the capture chain on 2 048 NV12 captures (tools/extract_nv12_probe.py), two-pass warp (default) and the conversion
# inside the warp kernel (CIMBAR_HIP_WARP_TWOPASS=0). Usage: tools/gpu_profile_nv12.sh <tag>  ->  gpurun_out/<tag>_nv12_{twopass,onepass}_kernel_stats.csv + _probe.txt
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/nv12_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in twopass onepass; do
	if [ $v = onepass ]; then export CIMBAR_HIP_WARP_TWOPASS=0; fi
	timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python $R/tools/extract_nv12_probe.py 2048 > $OUT/$v.log 2>&1
	grep '"captures"' $OUT/$v.log | tail -1 >> $R/gpurun_out/${TAG}_nv12_probe.txt
	python - "$OUT/$v" "$R/gpurun_out/${TAG}_nv12_${v}_kernel_stats.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        r = csv.reader(f)
        head = next(r)
        rows = [head] + [[x[0].replace("(anonymous namespace)::", "")[:110]] + x[1:] for x in r if "m68" in x[0] or "k_threshold" in x[0]]
with open(sys.argv[2], "w", newline="") as f:
    csv.writer(f).writerows(rows)
PY
done
rm -rf $OUT
