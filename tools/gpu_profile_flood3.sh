#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for the exact flood replay as the library ships it (k_flood3) on BASELINE configs[4]
# captures (tools/config5_bench.py: 256 and 1024 device-resident 1080p captures -> scan, extract, decode; every frame takes the exact replay).
# Kernel trace first, counters in their own passes (no --sys-trace etc. next to --pmc).   Usage: tools/gpu_profile_flood3.sh <tag>
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/flood3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/config5_bench.py"
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
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python - "$OUT" "${OUT}_pmc.json" <<'PY'
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)
out, dst = sys.argv[1:3]
acc = {}
for path in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            name = None
            for cand in ("k_flood3", "k_flood_wave", "k_warp_matrices", "k_warp", "k_scan_gray_blur_rows", "k_scan_gray_blur", "k_scan_otsu", "k_scan_stage_rows", "k_scan_stage1",
                         "k_scan_confirm", "k_scan_select", "k_scan_final", "k_scan_offsets", "k_threshold", "k_symbols", "k_colors", "k_rs", "k_frame_mid", "k_frame_end"):
                if cand in k:
                    name = cand
                    break
            if name is None:
                continue
            d = acc.setdefault(name + " grid=" + row.get("Grid_Size", "?"), {})
            e = d.setdefault(row["Counter_Name"], [0.0, 0])
            e[0] += float(row["Counter_Value"]); e[1] += 1
res = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} | {"dispatches": max(v[1] for v in d.values())} for k, d in acc.items()}
json.dump({"per_dispatch_average": res, "note": "tools/config5_bench.py under rocprofv3 --pmc (four passes): 256 and 1024 captures of 1920x1080; grid = threads per launch; FETCH_SIZE / WRITE_SIZE (own passes) in KiB (FETCH_SIZE counts 64 B per 128-B request on gfx950: x2 for bytes)"}, open(dst, "w"), indent=1)
print(json.dumps(res)[:3000])
PY
rm -rf $OUT
