#!/bin/bash
# tools/r5_fused_ab.sh: the bounce loop of a pass as one persistent kernel (pt_fused.h, PT_FUSED=2) against the launch chain at a 1/8 share,
# a 1/4 share, a 1/2 share and the full frame, with window sizes of 64 .. 512 entries per wave.  The fused kernel of ONE pixel chunk fills
# the chip with its persistent waves, so three chunk streams serialise (S3 rows); it wants the whole share as one pass (--streams 1).
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
CFGS=("chain PT_FUSED=0")
for c in ${CAPS:-64 128 256}; do CFGS+=("cap$c PT_FUSED=2 PT_FUSED_CAP=$c"); done
for W in ${WORLDS:-8 4 2 1}; do
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  for S in ${STREAMS:-3 1}; do
    echo "== simulate-world $W, streams $S"
    ROUNDS=${ROUNDS:-2} BENCH_ARGS="$B --steps 30 $A --streams $S" bash tools/r3_ab_env.sh "${CFGS[@]}" 2>&1 | tail -${#CFGS[@]}
  done
done
