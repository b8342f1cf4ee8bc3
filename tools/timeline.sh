#!/bin/bash
# tools/timeline.sh <tag> [bench args]: kernel timeline of the LAST frame of a short bench run (rocprofv3 --kernel-trace):
# per dispatch start offset, duration, queue; gaps between consecutive dispatches of a queue.  Output: gpurun_out/timeline_<tag>.txt
set -e
TAG=$1; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/tl_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules "$@" > "$OUT/log" 2>&1 || { tail -20 "$OUT/log"; exit 1; }
python3 tools/timeline.py "$OUT" > "$PWD/gpurun_out/timeline_$TAG.txt"
tail -1 "$OUT/log" >> "$PWD/gpurun_out/timeline_$TAG.txt"
rm -rf "$OUT"
