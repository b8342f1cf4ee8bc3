#!/bin/bash
# tools/r5_cp_sweep.sh: cost of a triangle test relative to a wide-node step in the SAH-optimal collapse (PT_BVH_CP, default 0.3), re-swept on the SAH hierarchy
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for wl in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 terrain1M_textured_1080p_4spp_d8; do
  echo "== $wl"
  ROUNDS=2 BENCH_ARGS="$B --steps 30 --workload $wl" bash tools/r3_ab_env.sh "cp03 PT_BVH_CP=0.3" "cp12 PT_BVH_CP=1.2" "cp2 PT_BVH_CP=2" "cp3 PT_BVH_CP=3" "cp5 PT_BVH_CP=5" "cp10 PT_BVH_CP=10" 2>&1 | tail -6
done
