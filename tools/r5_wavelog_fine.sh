#!/bin/bash
# tools/r5_wavelog_fine.sh: cycles of the traversal waves' OUTER loop parts (write-back, refill, steal round; -DPT_DEBUG_WAVELOG=2 variant `wlog2`)
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for W in 8 1; do
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  PT_FUSED=0 PT_LIB=$V/libptamd_wlog2.so PT_DEBUG_COUNTS=1 PT_WAVELOG=$PWD/gpurun_out/r5_wl2.bin python bench.py $B --steps 2 --warmup 2 $A > gpurun_out/r5_wl2.json 2> gpurun_out/r5_wl2.err || { tail -5 gpurun_out/r5_wl2.err; exit 1; }
  python tools/r5_wavelog.py gpurun_out/r5_wl2.bin --fine > gpurun_out/r5_wavelog_fine_w$W.txt
  rm -f gpurun_out/r5_wl2.bin
done
