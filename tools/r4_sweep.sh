#!/bin/bash
# tools/r4_sweep.sh NAME...: kernel-parameter variants against the shipped build on C3 and the stadium (one pt_render per frame)
V=$PWD/optixpathtracer_amd/variants
CFGS=("base X=1")
for n in "$@"; do CFGS+=("$n PT_LIB=$V/libptamd_$n.so"); done
for WL in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8; do
  echo "== $WL"
  ROUNDS=2 BENCH_ARGS="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30 --workload $WL" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -$((${#CFGS[@]}))
done
