#!/bin/bash
# tools/r5_fused_scenes.sh: the default policy (small synchronous frames as one fused pass) against the launch chain on the other scenes, 1/8 and 1/4 shares
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30"
for WL in ${WLS:-c2_cornell_1080p_4spp_d8 stadium1M_1080p_4spp_d8}; do
for W in 8 4; do
  echo "== $WL simulate-world $W"
  ROUNDS=2 BENCH_ARGS="$B --workload $WL --simulate-world $W" bash tools/r3_ab_env.sh "chain PT_FUSED=0" "fused PT_FUSED=1" 2>&1 | tail -2
done
done
