#!/bin/bash
# tools/r5_fused_policy.sh: PT_FUSED=1 (frames of at most PT_FUSED_MAX_PATHS paths: one pass, one fused kernel) against the default schedule
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30"
for W in ${WORLDS:-8 4 2}; do
  echo "== simulate-world $W"
  ROUNDS=${ROUNDS:-3} BENCH_ARGS="$B --simulate-world $W" bash tools/r3_ab_env.sh "chain PT_FUSED=0" "fused1 PT_FUSED=1 PT_FUSED_MAX_PATHS=5000000" 2>&1 | tail -2
done
