#!/bin/bash
# tools/r5_tail_ab.sh: the two stalls taken out of the stealing phase (PT8_STEAL_WSYNC: wave-level instead of workgroup barriers in the steal round;
# PT8_DEFER_STEAL: rays finished in the stealing phase are written back together) against the old kernel, chain and fused, at every share
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 30"
for W in ${WORLDS:-8 4 1}; do
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  echo "== simulate-world $W"
  ROUNDS=${ROUNDS:-3} BENCH_ARGS="$B $A" bash tools/r3_ab_env.sh "old_chain PT_FUSED=0 PT_LIB=$V/libptamd_old.so" "ws_chain PT_FUSED=0 PT_LIB=$V/libptamd_wsonly.so" "new_chain PT_FUSED=0" \
     "old_fused PT_FUSED=1 PT_LIB=$V/libptamd_old.so" "new_fused PT_FUSED=1" 2>&1 | tail -5
done
