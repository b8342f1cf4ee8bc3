#!/bin/bash
# tools/r5_fused_small.sh: window size of the fused loop by frame size (1/4 ... 1/32 of the 1080p frame: 2.07 M ... 0.26 M paths)
B="--no-cpu-baseline --no-isolated --no-extra-schedules --steps 40 --streams 1"
for W in ${WORLDS:-4 6 12 16 32}; do
  echo "== simulate-world $W"
  CFGS=("chain3 PT_FUSED=0")
  for c in ${CAPS:-64 128 192 256}; do CFGS+=("c$c PT_FUSED=2 PT_FUSED_CAP=$c"); done
  ROUNDS=2 BENCH_ARGS="$B --simulate-world $W" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -${#CFGS[@]}
done
