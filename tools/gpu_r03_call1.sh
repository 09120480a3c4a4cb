#!/bin/bash
# GPU call: the whole -m gpu suite, then the exact flood's kernel variants timed on BASELINE configs[4] captures and on shifted frames,
# then the cycle counters of the FLOOD_PROF build. Everything under its own timeout.
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -x -k "flood_verify or cpp_adapter or padded or gpu_flood" > $O/r03_t1a.log 2>&1; echo "tests(new) rc=$?"; tail -5 $O/r03_t1a.log
for v in default LV5 FLOOD2; do
  case $v in default) E="";; LV5) E="CIMBAR_HIP_FLOOD_LV6=0";; FLOOD2) E="CIMBAR_HIP_FLOOD3=0";; esac
  env $E timeout 300 python tools/config5_bench.py > $O/r03_c5_$v.log 2>&1; echo "== config5 $v"; grep -v Warning $O/r03_c5_$v.log | tail -3
  env $E CIMBAR_HIP_FLOOD_WAVE=0 timeout 300 python tools/flood_bench.py > $O/r03_fb_$v.log 2>&1; echo "== flood_bench $v"; tail -4 $O/r03_fb_$v.log
done
CIMBAR_HIP_LIB=$R/tools/_prof/libcimbar_hip_prof.so FLOOD_PROF_OUT=$O/r03_flood3_prof.json timeout 400 python tools/flood_prof.py > $O/r03_prof.log 2>&1; echo "== prof"; tail -6 $O/r03_prof.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 > $O/r03_t1.log 2>&1; echo "tests(all) rc=$?"; tail -8 $O/r03_t1.log
