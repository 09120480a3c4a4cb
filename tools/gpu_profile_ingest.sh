#!/bin/bash
# Runs on the GPU box (via gpurun): does the ingest ring overlap its host->device copies with the decode? rocprofv3 kernel + memory-copy trace
# of tools/ingest_trace.py, reduced to: busy time of the copy engine, busy time of the kernels, and the time both are busy at once.
# Usage: tools/gpu_profile_ingest.sh <tag>
TAG=${1:-run}
R=$PWD
OUT=$R/gpurun_out/ingest_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python $R/tools/ingest_trace.py > $OUT/trace.log 2>&1
tail -3 $OUT/trace.log
python - "$OUT/trace" "${OUT}_overlap.json" <<'PY'
import csv, glob, json, os, sys
csv.field_size_limit(1 << 30)
d, dst = sys.argv[1:3]
def spans(pattern, start, end, keep=lambda r: True):
    out = []
    for path in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if keep(r):
                    out.append((int(r[start]), int(r[end])))
    return sorted(out)
def union(sp):
    out = []
    for a, b in sp:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out
def length(u):
    return sum(b - a for a, b in u)
def intersect(u, v):
    i = j = 0
    tot = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b:
            tot += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return tot
kern = spans("*kernel_trace.csv", "Start_Timestamp", "End_Timestamp", lambda r: "m68" in r.get("Kernel_Name", ""))
h2d = spans("*memory_copy_trace.csv", "Start_Timestamp", "End_Timestamp", lambda r: "HOST_TO_DEVICE" in r.get("Direction", "") or "H2D" in r.get("Direction", ""))
big = [s for s in h2d if s[1] - s[0] > 500000]          # the 64-frame batch copies (201 MB each), not the small table uploads
# the three timed runs are the last 24 batch copies (8 batches of 64 frames each)
big = big[-24:]
t0, t1 = big[0][0], max(b for _, b in big)
ku = union([s for s in kern if s[1] > t0 and s[0] < t1])
cu = union(big)
res = {"window_ms": (t1 - t0) / 1e6, "h2d_busy_ms": length(cu) / 1e6, "kernel_busy_ms": length(ku) / 1e6, "both_busy_ms": intersect(ku, cu) / 1e6,
       "h2d_batch_copies": len(big), "h2d_bytes_per_copy": 64 * 1024 * 1024 * 3,
       "kernel_time_hidden_under_copies_frac": round(intersect(ku, cu) / max(length(ku), 1), 3),
       "note": "tools/ingest_trace.py: 3 x 512 page-locked frames through cimbar_ingest_run_raw, batches of 64, ring of 3"}
res["h2d_GBs_while_copying"] = round(len(big) * 64 * 1024 * 1024 * 3 / max(length(cu), 1), 2)
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT
