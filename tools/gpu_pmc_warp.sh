#!/bin/bash
# Runs on the GPU box (via gpurun): what bounds k_warp (X4), and what the XCD-aware tile order does to it. For CIMBAR_HIP_WARP_ORDER = 0 (launch order)
# and 1 (eight contiguous runs of tiles, one per XCD): the kernel trace, then HBM / L2 counters and the instruction mix in passes of their own
# (no --sys-trace etc. next to --pmc). 1 024 device-resident 1080p captures through scan -> extract -> decode.   Usage: tools/gpu_pmc_warp.sh <tag> [captures]
TAG=${1:-run}; N=${2:-1024}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for ORDER in 0 1; do
  OUT=$R/gpurun_out/warp_${TAG}_order$ORDER
  mkdir -p $OUT
  export CIMBAR_HIP_WARP_ORDER=$ORDER
  CMD="python $R/tools/dense_probe.py --worker $N 1"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1
  rm -rf $OUT/pmc_* $OUT/trace
  python - "$OUT" "$ORDER" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "_summary.json"))
for r in d["kernel_stats"]:
    if "k_warp" in r["kernel"]:
        print("order", sys.argv[2], "trace", r["kernel"][:50], "calls", r["calls"], "avg_us", round(r["avg_ns"] / 1e3, 1))
for k, v in d["pmc"].items():
    if "k_warp" in k and "matrices" not in k:
        print("order", sys.argv[2], "pmc", k[:50], {a: b for a, b in v.items() if not a.startswith("_")}, "grid", v.get("_grid"))
PY
done
