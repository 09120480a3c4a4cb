#!/bin/bash
# Runs on the GPU box (via gpurun): start / end timestamps of every kernel of the pipelined headline loop (tools/rs_probe.py), for a timeline
# of how the chain kernels of one batch interleave with the threshold kernel of the next.   Usage: tools/gpu_trace_pipeline.sh <tag>
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $R/tools/rs_probe.py 40 > $OUT/log.txt 2>&1
python - "$OUT/t" "${OUT}_kernel_trace.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f, open(sys.argv[2], "w", newline="") as g:
        r = csv.DictReader(f)
        w = csv.writer(g)
        w.writerow(["kernel", "queue", "start_ns", "end_ns"])
        for row in r:
            name = row.get("Kernel_Name", "")
            if "m68" not in name and "k_threshold" not in name:
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("m68::", "").split("(")[0]
            w.writerow([short[:40], row.get("Queue_Id", ""), row["Start_Timestamp"], row["End_Timestamp"]])
PY
rm -rf $OUT/t
wc -l ${OUT}_kernel_trace.csv
