#!/bin/bash
# tools/r5_fused_log.sh: per-wave log of the fused bounce-loop kernel (PT_DEBUG_WAVELOG variant `wlog`) at a 1/8 share, one pass (--streams 1)
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules --simulate-world ${W:-8} --streams 1"
for c in ${CAPS:-64 128}; do
  PT_FUSED=2 PT_FUSED_CAP=$c PT_LIB=$V/libptamd_wlog.so PT_DEBUG_COUNTS=1 PT_WAVELOG=$PWD/gpurun_out/r5_fusedlog.bin python bench.py $B --steps 2 --warmup 2 > gpurun_out/r5_fusedlog.json 2> gpurun_out/r5_fusedlog.err || { tail -5 gpurun_out/r5_fusedlog.err; exit 1; }
  echo "== cap $c"; python tools/r5_wavelog.py gpurun_out/r5_fusedlog.bin | tee gpurun_out/r5_fusedlog_c$c.txt
  rm -f gpurun_out/r5_fusedlog.bin
done
