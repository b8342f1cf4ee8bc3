#!/bin/bash
# tools/r5_node64_ab.sh: the one-line 64-byte node (variant n64, -DPT8_NODE64=1) against the 80-byte node: C3, stadium, a 1/8 share of C3
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-extra-schedules"
for cfg in "c3 --workload c3_terrain1M_1080p_4spp_d8" "stadium --workload stadium1M_1080p_4spp_d8" "share8 --workload c3_terrain1M_1080p_4spp_d8 --simulate-world 8 --no-isolated"; do
  set -- $cfg; name=$1; shift
  echo "== $name"
  ROUNDS=${ROUNDS:-2} BENCH_ARGS="$B --steps 30 $*" bash tools/r3_ab_env.sh "n80 X=1" "n64 PT_LIB=$V/libptamd_n64.so" 2>&1 | tail -2
done
