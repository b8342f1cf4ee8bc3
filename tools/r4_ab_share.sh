#!/bin/bash
# tools/r4_ab_share.sh NAME...: variants against the shipped build at a 1/8 share, a 1/4 share and the full frame (one pt_render per frame)
V=$PWD/optixpathtracer_amd/variants
CFGS=("base X=1")
for n in "$@"; do CFGS+=("$n PT_LIB=$V/libptamd_$n.so"); done
for W in 8 4 1; do
  echo "== simulate-world $W"
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  ROUNDS=2 BENCH_ARGS="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30 $A" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -$((${#CFGS[@]}))
done
