#!/bin/bash
# pipeline-depth sweep of the headline bench (distinct inputs): ms per step pipelined / K1 alone / ordinary call
for d in ${DEPTHS:-2 3 4}; do
  CIMBAR_HIP_PIPE_DEPTH=$d python bench.py --no-cpu-baseline --no-extras > /tmp/b_${d}_$$.json 2> /tmp/b_${d}_$$.err
  tail -c 300 /tmp/b_${d}_$$.err | tail -2
  python - "$d" /tmp/b_${d}_$$.json <<'PY'
import json, sys
d = sys.argv[1]
try:
    j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("depth", d, j["ms_per_step"], j["stage_ms"]["threshold"], j["no_pipeline"]["ms_per_step"])
except Exception as e:
    print("depth", d, "failed", e)
PY
done
