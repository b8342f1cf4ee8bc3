#!/bin/bash
# tools/r5_il_ab.sh: interleaved static assignment of rays to waves (PT8_INTERLEAVE), cross-wave stealing on the new base (static chunk =
# stealing phase from the first iteration), and both, at a 1/8 share, a 1/4 share and the full frame
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for W in 8 4 1; do
  echo "== simulate-world $W"
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  ROUNDS=${ROUNDS:-2} BENCH_ARGS="$B --steps 30 $A" bash tools/r3_ab_env.sh "base X=1" "il PT_LIB=$V/libptamd_il.so" "il32k PT_LIB=$V/libptamd_il32k.so" 2>&1 | tail -5
done
