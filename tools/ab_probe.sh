#!/bin/bash
# A/B of two builds of the library on one box, interleaved: CIMBAR_HIP_LIB selects the .so (decoder.load_library)
A=${A:-libcimbar_amd/libcimbar_hip.so}
B=${B:-libcimbar_amd/libcimbar_hip_nofuse.so}
for k in 1 2 3; do
  for lib in $A $B; do
    CIMBAR_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras > /tmp/ab_$$.json 2> /tmp/ab_$$.err
    python - "$lib" /tmp/ab_$$.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], "pipelined", j["ms_per_step"], "ordinary", j["no_pipeline"]["ms_per_step"], {k: v for k, v in j["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "failed", e, open(sys.argv[2].replace(".json", ".err")).read()[-300:])
PY
  done
done
