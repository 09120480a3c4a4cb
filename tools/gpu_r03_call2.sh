#!/bin/bash
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/r03_t2.log 2>&1; echo "tests(all) rc=$?"; tail -12 $O/r03_t2.log
for v in roles noroles; do
  case $v in roles) E="";; noroles) E="CIMBAR_HIP_FLOOD_ROLES=0";; esac
  env $E timeout 300 python tools/config5_bench.py > $O/r03b_c5_$v.log 2>&1; echo "== config5 $v"; grep -v Warning $O/r03b_c5_$v.log | tail -2
  env $E CIMBAR_HIP_FLOOD_WAVE=0 timeout 300 python tools/flood_bench.py > $O/r03b_fb_$v.log 2>&1; echo "== flood_bench $v"; tail -2 $O/r03b_fb_$v.log
done
timeout 900 python bench.py --steps 60 --warmup 6 > $O/bench_r03a.json 2> $O/bench_r03a.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_r03a.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
    e = d.get("extra", {})
    for k in ("config3_cell_errors", "config4_n1", "config5_extract", "config5_extract_1024", "ingest_png_device"):
        print(k, json.dumps(e.get(k))[:900])
    print("cpu", d.get("cpu_baseline"))
except Exception as ex:
    print("bench parse failed", ex)
PY
tail -5 $O/bench_r03a.err
bash tools/gpu_profile_flood3.sh r03a > $O/r03a_prof.log 2>&1; tail -3 $O/r03a_prof.log | cut -c1-1500
