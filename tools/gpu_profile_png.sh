#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two PMC passes of the device PNG decode (tools/png_bench.py, kernels only:
# 2048 level-1 frame PNGs with one stream per wavefront and the 8 KiB ring, 8192 with four streams per wavefront). Usage: tools/gpu_profile_png.sh <tag>
TAG=${1:-png}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/png_bench.py 2048 0 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o pmc$i -- $CMD > $OUT/pmc$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT > $OUT.log 2>&1; head -c 2500 $OUT.log; rm -rf $OUT
