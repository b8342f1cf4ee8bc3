#!/bin/bash
# tools/r5_shade_cus.sh: k_shade on CUs of its own.  PT_SHADE_CUS=N: the chunk chains' streams get a CU mask without
# N CUs (hipExtStreamCreateWithCUMask), every k_shade launch goes to an unmasked stream behind an event.  N=0 is the shipped schedule, N=-1…
# C3, stadium and a 1/8 share of C3; medians of 2 x 30 frames.  Output: gpurun_out/r5_shade_cus.log
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for cfg in "c3 --workload c3_terrain1M_1080p_4spp_d8" "stadium --workload stadium1M_1080p_4spp_d8" "share8 --workload c3_terrain1M_1080p_4spp_d8 --simulate-world 8"; do
  set -- $cfg; name=$1; shift
  echo "== $name"
  ROUNDS=2 BENCH_ARGS="$B --steps 30 $*" bash tools/r3_ab_env.sh "n0 PT_SHADE_CUS=0" "n16 PT_SHADE_CUS=16" "n32 PT_SHADE_CUS=32" "n48 PT_SHADE_CUS=48" "n64 PT_SHADE_CUS=64" 2>&1 | tail -5
done
