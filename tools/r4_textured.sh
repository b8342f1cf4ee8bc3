#!/bin/bash
# tools/r4_textured.sh <tag>  (GPU box, from the repo root): the textured 1 M-triangle workload beside the untextured C3 — bench lines and
# k_shade's traffic per dispatch (FETCH_SIZE / WRITE_SIZE in separate passes), kernel-trace stats of one chunk stream for both.
set -e
TAG=${1:-r4_tex}
OUT=$PWD/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in c3_terrain1M_1080p_4spp_d8 terrain1M_textured_1080p_4spp_d8; do
  python3 bench.py --workload $WL --steps 20 --warmup 3 ${R4_NOCPU:+--no-cpu-baseline} > "$OUT/line_$WL.json" 2> "$OUT/line_$WL.err"
  echo "bench $WL done"
  ARGS="bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1_$WL" -- python3 $ARGS --streams 1 > "$OUT/stats1_$WL.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$WL" -- python3 $ARGS > "$OUT/fetch_$WL.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$WL" -- python3 $ARGS > "$OUT/write_$WL.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --output-format csv -d "$OUT/lat_$WL" -- python3 $ARGS > "$OUT/lat_$WL.log" 2>&1
  echo "profile $WL done"
done
python3 tools/r4_textured_summary.py "$OUT" "$TAG"
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete || true
find "$OUT" -name "*counter_collection.csv" -size +2M -delete || true
