#!/bin/bash
# tools/r5_fused_stadium.sh: window size of the fused bounce loop on the stadium (heavy-tailed ray costs) at a 1/8 share
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30 --workload stadium1M_1080p_4spp_d8 --simulate-world 8"
ROUNDS=2 BENCH_ARGS="$B" bash tools/r3_ab_env.sh "chain PT_FUSED=0" "c64 PT_FUSED=1 PT_FUSED_CAP=64" "c96 PT_FUSED=1 PT_FUSED_CAP=96" "c128 PT_FUSED=1" "c256 PT_FUSED=1 PT_FUSED_CAP=256" "c128g3072 PT_FUSED=1 PT_FUSED_GRID=3072" 2>&1 | tail -6
