#!/bin/bash
# tools/r4_gate.sh: experiment — the chain of pixel chunk c+1 of a synchronous frame starts behind the k-th launch of chunk c (PT_CHUNK_GATE=k; 0 = host enqueue order only)
GATES=${GATES:-"0 1 2 3 4 5"}
for WL in c3_terrain1M_1080p_4spp_d8 stadium1M_1080p_4spp_d8 ${EXTRA_WL}; do
  echo "== $WL"
  CFGS=()
  for g in $GATES; do CFGS+=("g${g}_$WL PT_CHUNK_GATE=$g"); done
  ROUNDS=${ROUNDS:-2} BENCH_ARGS="--no-cpu-baseline --no-isolated --no-extra-schedules --workload $WL" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -$(( $(echo $GATES | wc -w) + 1 ))
done
