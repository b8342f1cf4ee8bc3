#!/bin/bash
# tools/r5_fused_tune.sh: window size, grid and occupancy of the fused bounce-loop kernel at a 1/8 share (one pass: --streams 1)
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30 --simulate-world ${W:-8}"
F="PT_FUSED=2"
ROUNDS=2 BENCH_ARGS="$B --streams 1" bash tools/r3_ab_env.sh "c96 $F PT_FUSED_CAP=96" "c128 $F PT_FUSED_CAP=128" "c160 $F PT_FUSED_CAP=160" \
  "w4c64 $F PT_FUSED_CAP=64 PT_LIB=$V/libptamd_fw4.so" "w4c128 $F PT_FUSED_CAP=128 PT_LIB=$V/libptamd_fw4.so" "w4c192 $F PT_FUSED_CAP=192 PT_LIB=$V/libptamd_fw4.so" \
  "g4096c128 $F PT_FUSED_CAP=128 PT_FUSED_GRID=4096" "g3072c192 $F PT_FUSED_CAP=192 PT_FUSED_GRID=3072" "g2048c256 $F PT_FUSED_CAP=256 PT_FUSED_GRID=2048" 2>&1 | tail -9
