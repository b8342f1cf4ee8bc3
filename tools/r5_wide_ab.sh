#!/bin/bash
# tools/r5_wide_ab.sh [variants...]: wide start of small traversal launches (PT8_WIDE_MIN rays per wave, PT8_STEAL_PERIOD) against the
# shipped build at a 1/8 share, a 1/4 share and the full frame
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
CFGS=("base X=1")
for n in "$@"; do CFGS+=("$n PT_LIB=$V/libptamd_$n.so"); done
for W in 8 4 1; do
  echo "== simulate-world $W"
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  ROUNDS=${ROUNDS:-2} BENCH_ARGS="$B --steps 30 $A" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -${#CFGS[@]}
done
