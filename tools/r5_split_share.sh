#!/bin/bash
# tools/r5_split_share.sh: the shadow schedules (0 unified launches, 1 shadow launches on a second stream, 2 asynchronous shadow records) at a 1/8 and a 1/4 share
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
for W in 8 4; do
  echo "== simulate-world $W"
  ROUNDS=2 BENCH_ARGS="$B --steps 30 --simulate-world $W" bash tools/r3_ab_env.sh "s0 X=1" 2>&1 | tail -1
  for S in 1 2; do
    ROUNDS=2 BENCH_ARGS="$B --steps 30 --simulate-world $W --split-shadow $S" bash tools/r3_ab_env.sh "s$S X=1" 2>&1 | tail -1
  done
done
