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
# every dispatch of the run that is one of ours and takes more than 20 us, relative to the first one (the warm-up batch comes first, then the timed
# batch, then the extract-only batch)
import re
t0 = rows[0][0]
for s, e, name, grid, q in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0][:40]
    if "k_" in name and e - s > 20000:
        print(f"{(s - t0) / 1e6:9.2f} -> {(e - t0) / 1e6:9.2f} ms ({(e - s) / 1e6:7.3f})  {short:42s} grid {grid:>9s} queue {q}")
PY
tail -2 $OUT/trace.log
rm -rf $OUT
