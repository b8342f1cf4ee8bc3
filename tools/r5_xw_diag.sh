#!/bin/bash
# NEEDS tools/patches/r5_xw_stealing.patch applied (git apply; the cross-wave code was taken out of the product after these runs).
# tools/r5_xw_diag.sh: where the time of a cross-wave stealing launch goes at a 1/8 share — the XW kernel variant alone (e1; e1w5: at five
# waves per SIMD), + the report atomic (e2), + lingering without donations (poll0), everything (xw) — and the per-wave log of poll0
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
ROUNDS=2 BENCH_ARGS="$B --steps 30 --simulate-world 8" bash tools/r3_ab_env.sh "off PT_XW=0" "e1 PT_XW=1 PT_LIB=$V/libptamd_xwe1.so" "e1w5 PT_XW=1 PT_LIB=$V/libptamd_xwe1w5.so" "e2 PT_XW=1 PT_LIB=$V/libptamd_xwe2.so" "poll0 PT_XW=1 PT_LIB=$V/libptamd_xwp0.so" "xw PT_XW=1" 2>&1 | tail -6
PT_XW=1 PT_LIB=$V/libptamd_xwp0log.so PT_DEBUG_COUNTS=1 PT_WAVELOG=$PWD/gpurun_out/r5_xwlog_w8.bin python bench.py $B --steps 2 --warmup 2 --simulate-world 8 > gpurun_out/r5_xwlog_w8.json 2> gpurun_out/r5_xwlog_w8.err
python tools/r5_wavelog.py gpurun_out/r5_xwlog_w8.bin > gpurun_out/r5_xwp0log_w8.txt
rm -f gpurun_out/r5_xwlog_w8.bin
