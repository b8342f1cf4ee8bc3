#!/bin/bash
# tools/r5_sah_ab2.sh: SAH hierarchy variants (bins over the node box / over the centroid bounds / 32 bins) against the forced LBVH, interleaved rounds:
# C3, stadium, a 1/8 share of C3, the 9.5 M-triangle terrain
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for cfg in "c3 --steps 30 --workload c3_terrain1M_1080p_4spp_d8" "stadium --steps 30 --workload stadium1M_1080p_4spp_d8" "share8 --steps 30 --workload c3_terrain1M_1080p_4spp_d8 --simulate-world 8" "t10M --steps 20 --workload terrain10M_1080p_4spp_d8"; do
  set -- $cfg; name=$1; shift
  echo "== $name"
  ROUNDS=3 BENCH_ARGS="$B $*" bash tools/r3_ab_env.sh "lbvh PT_BVH_BUILDER=lbvh" "sahbox PT_BVH_BUILDER=sah PT_LIB=$V/libptamd_sahc0.so" "sahcen PT_BVH_BUILDER=sah PT_LIB=$V/libptamd_sahc1.so" "sahcen32 PT_BVH_BUILDER=sah PT_LIB=$V/libptamd_sahc1b32.so" 2>&1 | tail -4
done
