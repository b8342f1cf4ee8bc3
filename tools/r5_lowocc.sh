#!/bin/bash
# tools/r5_lowocc.sh: the traversal kernels compiled for 2 / 3 waves per SIMD (more registers: no spills, freer schedule) at small shares, where
# occupancy is not what limits (about one wave per SIMD has work)
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30"
for W in 8 4; do
  echo "== simulate-world $W"
  ROUNDS=2 BENCH_ARGS="$B --simulate-world $W" bash tools/r3_ab_env.sh "chain5 PT_FUSED=0" "chain3 PT_FUSED=0 PT_LIB=$V/libptamd_w3.so" "chain2 PT_FUSED=0 PT_LIB=$V/libptamd_w2.so" \
     "fused5 PT_FUSED=1" "fused3 PT_FUSED=1 PT_LIB=$V/libptamd_w3.so" "fused2 PT_FUSED=1 PT_LIB=$V/libptamd_w2.so" 2>&1 | tail -6
done
