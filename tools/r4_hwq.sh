#!/bin/bash
# tools/r4_hwq.sh: more hardware queues for the HIP runtime (GPU_MAX_HW_QUEUES, default 4) and more pixel-chunk streams: 1/8 share and full frame
mkdir -p gpurun_out/ab
for W in 8 1; do
for Q in 4 8; do
for S in 3 4 5 6; do
  if [ $W = 1 ]; then A=""; else A="--simulate-world $W"; fi
  for r in 1 2; do
    GPU_MAX_HW_QUEUES=$Q python bench.py --no-cpu-baseline --no-isolated --no-extra-schedules --steps 40 --warmup 5 $A --streams $S > gpurun_out/ab/hq.json 2> gpurun_out/ab/hq.err || { tail -3 gpurun_out/ab/hq.err; exit 1; }
    python - "world $W hwq $Q streams $S" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab/hq.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:30s} ms {d['ms_per_step']:.3f}  median {d['step_ms']['median']:.3f}", flush=True)
PY
  done
done
done
done
