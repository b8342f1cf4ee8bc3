#!/bin/bash
# tools/r5_fused_params.sh: the traversal loop's parameters (refill threshold, steal period, triangle-step bias) under the fused bounce loop, 1/8 and 1/4 shares
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30"
for W in 8 4; do
  echo "== simulate-world $W"
  CFGS=("base X=1")
  for n in rf32 rf52 sp2 sp5 tb1 tb3; do CFGS+=("$n PT_LIB=$V/libptamd_$n.so"); done
  ROUNDS=2 BENCH_ARGS="$B --simulate-world $W" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -7
done
