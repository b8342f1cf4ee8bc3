#!/bin/bash
# tools/r4_share_streams.sh: a 1/8 share of C3 (one pt_render per frame) against the number of pixel-chunk streams and the path-slot budget (= chunk size)
mkdir -p gpurun_out/ab
for cfg in "--streams 1" "--streams 2" "--streams 3" "--streams 4" "--streams 6" "--streams 3 --max-paths 262144" "--streams 4 --max-paths 262144" "--streams 6 --max-paths 174763" "--streams 1 --max-paths 2000000" "--streams 2 --max-paths 524288"; do
  for r in 1 2; do
    python bench.py --no-cpu-baseline --no-isolated --no-extra-schedules --steps 40 --warmup 5 --simulate-world 8 $cfg > gpurun_out/ab/ss.json 2> gpurun_out/ab/ss.err || { tail -3 gpurun_out/ab/ss.err; exit 1; }
    python - "$cfg" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab/ss.json').read().strip().splitlines()[-1])
print(f"{sys.argv[1]:40s} ms {d['ms_per_step']:.3f}  median {d['step_ms']['median']:.3f}", flush=True)
PY
  done
done
