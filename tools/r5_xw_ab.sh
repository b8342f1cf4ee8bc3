#!/bin/bash
# NEEDS tools/patches/r5_xw_stealing.patch applied (git apply; the cross-wave code was taken out of the product after these runs).
# tools/r5_xw_ab.sh [ROUNDS]: cross-wave stealing A/B (PT_XW=0 off / 1 small passes / 2 every launch) at a 1/8 share, a 1/4 share and the full frame
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
R=${1:-2}
for W in 8 4 1; do
  echo "== simulate-world $W"
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  ROUNDS=$R BENCH_ARGS="$B --steps 30 $A" bash tools/r3_ab_env.sh "xw0 PT_XW=0" "xw1 PT_XW=1" "xw2 PT_XW=2" 2>&1 | tail -3
done
