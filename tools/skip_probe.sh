#!/bin/bash
# which chain kernels cost the pipelined step its time? CIMBAR_HIP_DEBUG_SKIP masks: 1 symbols, 2 flood, 4 rs<4>, 8 frame_mid, 16 colors, 32 rs<2>, 64 frame_end
# The switch only exists in the probe build (python -m libcimbar_amd.build --probes, on the CPU box before gpurun); bench.py --probe-run prints the
# timing fields only (no "metric" / "value": a line made with kernels dropped is not a result).
LIB=$PWD/libcimbar_amd/variants/libcimbar_hip_probes.so
[ -f $LIB ] || { echo "build the probe library first: python -m libcimbar_amd.build --probes"; exit 1; }
export CIMBAR_HIP_LIB=$LIB
for m in ${MASKS:-0 127 1 2 4 8 16 32 64 0}; do
  CIMBAR_HIP_DEBUG_SKIP=$m python bench.py --probe-run --no-cpu-baseline --no-extras --steps 100 > /tmp/s_$$.json 2> /tmp/s_$$.err
  python - "$m" /tmp/s_$$.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("skip", sys.argv[1], "pipelined", j["ms_per_step"], "K1", j["stage_ms"]["threshold"], "ordinary", j["no_pipeline"]["ms_per_step"], {k: v for k, v in j["stage_ms"].items() if k != "threshold"})
except Exception as e:
    print("skip", sys.argv[1], "failed", e)
PY
done
