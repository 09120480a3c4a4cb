#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for the extract stage (X1 blur/histogram, X2 Otsu, the anchor-scan kernels, X4 warp) on
# 256 device-resident 1080p captures (tools/extract_bench.py = bench.py's config5_extract row). Kernel trace first, HBM counters in their own pass.
# Usage: tools/gpu_profile_extract.sh <tag>
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/extract_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/extract_bench.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python - "$OUT/trace" "${OUT}_kernel_stats.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        r = csv.reader(f)
        head = next(r)
        rows = [head] + [x for x in r if "cimbar" in x[0] or "m68" in x[0] or "m67" in x[0]]
with open(sys.argv[2], "w", newline="") as f:
    csv.writer(f).writerows(rows)
PY
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $OUT/pmc -o p -- $CMD > $OUT/pmc.log 2>&1
python - "$OUT" "${OUT}_pmc.json" <<'PY'
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)
out, dst = sys.argv[1:3]
acc = {}
for path in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "k_scan_gray_blur" not in k and "k_warp" not in k:
                continue
            name = "k_warp_matrices" if "k_warp_matrices" in k else ("k_warp" if "k_warp" in k else "k_scan_gray_blur")
            grid = row.get("Grid_Size", "")
            d = acc.setdefault(name + " grid=" + grid, {})
            e = d.setdefault(row["Counter_Name"], [0.0, 0])
            e[0] += float(row["Counter_Value"]); e[1] += 1
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide streaming read at 64 bytes
# (MI355X_MICROARCH.md, HBM section), so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the same correction bench.py applies to K1
res = {}
for k, d in acc.items():
    r = {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()}
    r["dispatches"] = max(v[1] for v in d.values())
    r["hbm_bytes_corrected"] = int((2.0 * r.get("FETCH_SIZE", 0) + r.get("WRITE_SIZE", 0)) * 1024)
    res[k] = r
json.dump({"per_dispatch_average_KiB": res, "note": "tools/extract_bench.py: 256 captures of 1920x1080 per dispatch (plus warm-up dispatches on the same batch)"}, open(dst, "w"), indent=1)
print(json.dumps(res)[:2000])
PY
rm -rf $OUT
