#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel-trace stats + two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage: tools/profile.sh <tag> [bench args...]
set -e
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/bench_write.log" 2>&1
python3 tools/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary_$TAG.md"
# keep only the small artefacts (the per-dispatch CSVs are tens of MB)
find "$OUT" -name "*_kernel_trace.csv" -size +2M -delete || true
find "$OUT" -name "*counter_collection.csv" -size +2M -delete || true
ls -la "$OUT"
