#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 evidence for the two flood kernels on a batch of rigidly shifted frames (every frame flagged).
#   default library   -> k_flood_wave settles them (batch-parallel, certified)
#   CIMBAR_HIP_FLOOD_WAVE=0 -> k_flood (exact replay) takes all of them
# Usage: tools/gpu_profile_flood.sh <tag>     (counters in their own passes, no --sys-trace etc. next to --pmc)
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/flood_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/flood_bench.py"
for mode in wave exact; do
  if [ $mode = exact ]; then export CIMBAR_HIP_FLOOD_WAVE=0; else unset CIMBAR_HIP_FLOOD_WAVE; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- $CMD > $OUT/${mode}_trace.log 2>&1
  python - "$OUT/trace_$mode" "${OUT}_${mode}_kernel_stats.csv" <<'PY'
import csv, glob, os, sys
csv.field_size_limit(1 << 30)
rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
        r = csv.reader(f)
        head = next(r)
        rows = [head] + [x for x in r if "k_flood" in x[0] or "k_front" in x[0] or "k_symbols" in x[0] or "k_threshold" in x[0]]
with open(sys.argv[2], "w", newline="") as f:
    csv.writer(f).writerows(rows)
PY
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${mode}_$i -o p -- $CMD > $OUT/${mode}_pmc$i.log 2>&1
  done
  python - "$OUT" "$mode" "${OUT}_${mode}_pmc.json" <<'PY'
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)
out, mode, dst = sys.argv[1:4]
acc = {}
for path in glob.glob(os.path.join(out, f"pmc_{mode}_*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if "k_flood" not in k:
                continue
            name = "k_flood_wave" if "k_flood_wave" in k else "k_flood"
            d = acc.setdefault(name, {})
            c = row["Counter_Name"]
            v = float(row["Counter_Value"])
            e = d.setdefault(c, [0.0, 0])
            e[0] += v; e[1] += 1
res = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} for k, d in acc.items()}
json.dump({"mode": mode, "per_dispatch_average": res, "note": "tools/flood_bench.py: batches of 1, 16, 256 and 1024 frames shifted by (2,1); averages over all dispatches of the kernel"}, open(dst, "w"), indent=1)
print(json.dumps(res)[:1500])
PY
done
rm -rf $OUT
