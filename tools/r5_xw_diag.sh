#!/bin/bash
# tools/r5_xw_diag.sh: where the time of a cross-wave stealing launch goes — variants (poll never / every 4th iteration / assumed hungry)
# against PT_XW=0 at a 1/8 share, and the per-wave log of the XW build.  Output: gpurun_out/r5_xw_diag.log, gpurun_out/r5_xwlog_w8.txt
V=$PWD/optixpathtracer_amd/variants
B="--no-cpu-baseline --no-isolated --no-extra-schedules"
ROUNDS=1 BENCH_ARGS="$B --steps 30 --simulate-world 8" bash tools/r3_ab_env.sh "off PT_XW=0" "poll1 PT_XW=1" "poll0 PT_LIB=$V/libptamd_xwp0.so" "poll4 PT_LIB=$V/libptamd_xwp4.so" "pollm PT_LIB=$V/libptamd_xwpm.so" 2>&1 | tail -5
PT_LIB=$V/libptamd_xwlog.so PT_DEBUG_COUNTS=1 PT_WAVELOG=$PWD/gpurun_out/r5_xwlog_w8.bin python bench.py $B --steps 2 --warmup 2 --simulate-world 8 > gpurun_out/r5_xwlog_w8.json 2> gpurun_out/r5_xwlog_w8.err
python tools/r5_wavelog.py gpurun_out/r5_xwlog_w8.bin > gpurun_out/r5_xwlog_w8.txt
rm -f gpurun_out/r5_xwlog_w8.bin
