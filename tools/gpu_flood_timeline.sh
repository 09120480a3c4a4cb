#!/bin/bash
# start / end of every flood dispatch of one BASELINE configs[4] batch (1024 captures), relative to the first kernel of the batch
TAG=${1:-tl}
R=$PWD
OUT=$R/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python $R/tools/flood_timeline.py > $OUT/trace.log 2>&1
python - "$OUT/trace" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Queue_Id", "?")))
rows.sort()
# the last batch: from the last k_scan_otsu on
last = max(i for i, r in enumerate(rows) if "k_scan_stage1" in r[2])
t0 = rows[last][0]
for s, e, name, grid, q in rows[last:]:
    short = name.split("(")[0].split("::")[-1][:34]
    if "k_" in name:
        print(f"{(s - t0) / 1e6:8.2f} -> {(e - t0) / 1e6:8.2f} ms  {short:36s} grid {grid:>9s} queue {q}")
PY
tail -2 $OUT/trace.log
rm -rf $OUT
