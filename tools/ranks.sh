#!/bin/bash
# tools/ranks.sh N: frame time of every rank's share of an N-way partition, rendered one after the other on this GPU
cd "$(dirname "$0")/.."
N=${1:-8}
for r in $(seq 0 $((N-1))); do
  timeout -k 10 100 python bench.py --steps 20 --warmup 5 --simulate-world $N --simulate-rank $r --no-cpu-baseline --no-isolated 2>/dev/null > /tmp/rk.json
  python - $r <<'PY'
import json,sys
d=json.loads(open("/tmp/rk.json").read().strip().splitlines()[-1]); print("rank", sys.argv[1], d["ms_per_step"], "ms", d["rays_per_frame"], "rays", flush=True)
PY
done
