#!/bin/bash
# tools/r4_tiles.sh: a 1/8 share of C3 with larger tiles of the interleaved partition (a rank's pixels closer together), three of the eight ranks each
for T in "64 16" "128 64" "240 136" "480 272" "960 544"; do
  echo "tile $T"
  for RK in 0 3 6; do
    python bench.py --simulate-world 8 --simulate-rank $RK --tile $T --steps 30 --warmup 5 --no-cpu-baseline --no-isolated --no-extra-schedules 2>/dev/null > gpurun_out/tl.json
    python - $RK <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tl.json").read().strip().splitlines()[-1])
print("  rank", sys.argv[1], d["ms_per_step"], d["step_ms"]["median"], d["rays_per_frame"])
PY
  done
done
